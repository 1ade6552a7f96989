// Implicit-GEMM convolution family on fp32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
//   tg_conv_fwd    y  = act((conv(x (.) m, W) + b) * ratio)         pconv.py:27-30,43 / nn.Conv2d
//   tg_conv_dgrad  dx = convT(dy, W) (.) m                            autograd of the above
//   tg_conv_wgrad  dW = sum_pix dy (x) (x (.) m),  db = sum_pix dy
//
// Layout: activations NHWC, weights [Cout][kh][kw][Cin].  One 256-thread workgroup (4 waves of 64)
// owns a BM x BN output tile; K is walked in 32-deep steps through double-buffered LDS; each wave
// holds WM x WN accumulator tiles of 32x32 (16 VGPRs each).  LDS images are [row][k] with a 36-float
// row pitch so that the ds_read_b128 operand fetch (4 consecutive k per lane) is bank-conflict free;
// MFMA number e of a k-group consumes element e of that fetch on both operands, i.e. k = 8g+4h+e
// for lane half h -- a permutation of the k order, which a sum over k does not care about.
#include <atomic>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

// ------------------------------------------------------------------------------------------------
// optional per-launch timing of the MFMA kernels (bench.py's roofline figures): when enabled, every
// igemm / wgrad main-kernel launch is bracketed by hipEvents ON THE LAUNCH STREAM and tagged with
// its algorithmic FLOPs (2*M*N*K of the un-padded problem) and algorithmic bytes.
// ------------------------------------------------------------------------------------------------
struct ProfRec {
    hipEvent_t a, b;
    int kind;  // 0 = igemm (fwd/dgrad), 1 = wgrad
    double flops, bytes;
    int M, N, K, C, splits, cfg;
    char tag[32];
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static thread_local char g_prof_tag[32] = "";      // layer label of the launches that follow (tg_prof_tag)

struct ProfScope {
    bool on = false;
    ProfRec r{};
    hipStream_t s;
    ProfScope(hipStream_t st, int kind, double flops, double bytes, int M = 0, int N = 0, int K = 0, int C = 0, int splits = 1,
              int cfg = 0)
        : s(st) {
        if (!g_prof_on) return;
        on = true;
        r.kind = kind; r.flops = flops; r.bytes = bytes;
        r.M = M; r.N = N; r.K = K; r.C = C; r.splits = splits; r.cfg = cfg;
        memcpy(r.tag, g_prof_tag, sizeof(r.tag));
        (void)hipEventCreate(&r.a);
        (void)hipEventCreate(&r.b);
        (void)hipEventRecord(r.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(r);
    }
};

// algorithmic-FLOP scale of the launches issued on behalf of a rewritten problem (5x5 stride-2 layers run as 3x3 over a
// space-to-depth input: 25 of its 36 taps are real work), so that the profile tags count the ORIGINAL convolution
static thread_local double g_alg_scale = 1.0;
struct AlgScale {
    double prev;
    explicit AlgScale(double sc) : prev(g_alg_scale) { g_alg_scale = sc; }
    ~AlgScale() { g_alg_scale = prev; }
};

extern "C" int tg_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return TG_OK;
}
// Labels the profiled launches that follow on this thread (e.g. "dec1.fwd"); bench.py's per-layer roofline lines use it.
extern "C" int tg_prof_tag(const char* tag) {
    strncpy(g_prof_tag, tag ? tag : "", sizeof(g_prof_tag) - 1);
    g_prof_tag[sizeof(g_prof_tag) - 1] = 0;
    return TG_OK;
}
// Writes one CSV row per recorded launch (kind,cfg,M,N,K,C,splits,ms,gflop,alg_mb,tag) without consuming the records.
extern "C" int tg_prof_dump(const char* path) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    FILE* f = fopen(path, "w");
    if (!f) { tg_set_error("tg_prof_dump: cannot open %s", path); return TG_ERR_ARG; }
    fprintf(f, "kind,cfg,M,N,K,C,splits,ms,gflop,alg_mb,tag\n");
    for (auto& r : g_prof) {
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%.5f,%.4f,%.3f,%s\n", r.kind, r.cfg, r.M, r.N, r.K, r.C, r.splits, t, r.flops / 1e9,
                r.bytes / 1e6, r.tag);
    }
    fclose(f);
    return TG_OK;
}
// Synchronises the recorded events (host-blocking; never called inside a timed region) and returns the
// totals for `kind`; the records of that kind are consumed.
extern "C" int tg_prof_summary(int kind, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0, fl = 0, by = 0;
    int64_t n = 0;
    std::vector<ProfRec> keep;
    for (auto& r : g_prof) {
        if (r.kind != kind) { keep.push_back(r); continue; }
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        ms += t; fl += r.flops; by += r.bytes; ++n;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof.swap(keep);
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return TG_OK;
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad: gathered-row A operand, dense K-contiguous B operand
// ------------------------------------------------------------------------------------------------
#include "igemm_params.h"

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: opt in once per (kernel, device) --
// a process that drives a second GPU would otherwise launch the >64 KB-LDS kernels there without it.  Thread-safe.
struct LdsOptIn {
    std::mutex mu;
    bool done[64] = {};
};
static int lds_opt_in(LdsOptIn& st, const void* kern, size_t lds, const char* who) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(st.mu);
    if (dev >= 0 && dev < 64 && st.done[dev]) return TG_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        tg_set_error("%s: hipFuncSetAttribute(%zu) failed: %s", who, lds, hipGetErrorString(e));
        return TG_ERR_LAUNCH;
    }
    if (dev >= 0 && dev < 64) st.done[dev] = true;
    return TG_OK;
}

// CUs left free by the one-workgroup-per-CU Winograd kernels (tg_set_cu_reserve).  Those workgroups own their CU for the
// whole launch (141 KB LDS, 512 threads): a concurrent stream's small kernels -- RCCL's reduction kernels in a
// data-parallel run, whose gradient all-reduce is meant to run underneath the next kernels -- would otherwise queue behind
// a full launch.  248 of 256 workgroups dealt round-robin over the 8 XCDs leave one CU per XCD free.
// The reserve sizes the GRID of the persistent kernels only (they loop over their work items, so any grid is correct).
// Split-K plans -- which fix the summation order, i.e. the bits of the result -- are always made for WINO_PLAN_CUS, so a
// data-parallel run with a reserve produces exactly the numbers of the single-GPU path.  The value is read by launches on
// any host thread / stream: atomic.
constexpr int WINO_PLAN_CUS = 256;
static std::atomic<int> g_cu_reserve{0};
static int wino_cus() {
    const int c = 256 - g_cu_reserve.load(std::memory_order_relaxed);
    return c < 8 ? 8 : c;
}
extern "C" int tg_set_cu_reserve(int cus) {
    TG_REQUIRE(cus >= 0 && cus <= 128, "tg_set_cu_reserve: %d out of range [0,128]", cus);
    g_cu_reserve.store(cus, std::memory_order_relaxed);
    return TG_OK;
}

// Work-stealing item queues of the persistent Winograd kernels (WinoWork in wino.inc; tg_set_work_stealing).  A launch needs a
// small block of zeroed counters that nothing else touches while it runs: kernels on one stream never overlap, so every
// (device, stream) gets ONE block for good, handed out from a pool that is allocated (hipMalloc + hipMemset: synchronising
// calls) at the first request on a device -- never while that stream is capturing a graph (the launch then keeps the static
// distribution).  The kernels leave the block zeroed (the last workgroup to finish resets it).
static std::atomic<int> g_work_stealing{0};
extern "C" int tg_set_work_stealing(int mode) {
    TG_REQUIRE(mode >= 0 && mode <= 2, "tg_set_work_stealing: mode %d not in {0, 1, 2}", mode);
    g_work_stealing.store(mode, std::memory_order_relaxed);
    return TG_OK;
}
constexpr int WQ_HOST_Z = 16, WQ_HOST_INTS = 8 * WQ_HOST_Z + 8, WQ_POOL_BLOCKS = 64;
static int* wino_queue_block(hipStream_t s, int splits, int total_work) {
    static const int env_mode = getenv("TG_WORK_STEALING") ? atoi(getenv("TG_WORK_STEALING")) : 0;     // A/B on one GPU
    const int set_mode = g_work_stealing.load(std::memory_order_relaxed);
    const int mode = set_mode ? set_mode : env_mode;
    if (mode <= 0 || splits > WQ_HOST_Z) return nullptr;
    // mode 1: launches with fewer than two items per workgroup keep the static walk -- a late workgroup holds at most one
    // item there either way, and the queue only costs (profiles/r03_ws_contention.txt: dec4 forward).  Mode 2: every launch.
    if (mode == 1 && total_work < 2 * wino_cus()) return nullptr;
    static std::mutex mu;
    static std::vector<std::pair<std::pair<int, hipStream_t>, int*>> blocks;
    static int* pools[64] = {};
    static int used[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& b : blocks)
        if (b.first.first == dev && b.first.second == s) return b.second;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
        if (!pools[dev]) return nullptr;                 // cannot allocate inside a capture
    }
    if (!pools[dev]) {
        if (hipMalloc(reinterpret_cast<void**>(&pools[dev]), (size_t)WQ_POOL_BLOCKS * WQ_HOST_INTS * sizeof(int)) != hipSuccess) { pools[dev] = nullptr; return nullptr; }
        if (hipMemset(pools[dev], 0, (size_t)WQ_POOL_BLOCKS * WQ_HOST_INTS * sizeof(int)) != hipSuccess) return nullptr;
    }
    if (used[dev] >= WQ_POOL_BLOCKS) return nullptr;     // more streams than blocks: static distribution
    int* blk = pools[dev] + (size_t)used[dev]++ * WQ_HOST_INTS;
    blocks.push_back({{dev, s}, blk});
    return blk;
}

// Split-K factor with wave quantisation in mind: `slots` workgroups are resident at once (CUs x occupancy); a grid
// of 1.3 x slots long-running workgroups takes as long as 2 x slots.  Pick the smallest split count whose grid
// fills at least one round and wastes <= 8 % of its last round, else the most efficient one.
static int choose_splits(long tiles, int max_splits, int slots) {
    if (max_splits < 1) max_splits = 1;
    if (tiles >= 4L * slots) return 1;                 // many rounds already: the tail is small
    int best = 1;
    double best_eff = 0.0;
    for (int sp = 1; sp <= max_splits; ++sp) {
        const long blocks = tiles * sp;
        const long rounds = (blocks + slots - 1) / slots;
        const double eff = (double)blocks / ((double)rounds * slots);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = sp; }
        if (blocks >= slots && eff >= 0.92) return sp;
        if (blocks >= 6L * slots) break;
    }
    return best;
}


// The same decision when the length of the K walk is known (`ksteps` steps in all): every workgroup walks
// ceil(ksteps / splits) steps plus a fixed prologue / epilogue worth ~4 steps, `rounds` such walks follow each other, so the
// launch costs rounds x (ceil(ksteps / splits) + 4) step times -- minimise that; ties go to fewer splits (fewer slabs to
// write and reduce).  (choose_splits insists on filling a whole round first: 3 tiles x 85 splits = 255 workgroups on 256
// CUs lost to 158 splits = 1.85 rounds, 7 % slower with twice the slabs.)
// `slab_steps`: what one more (tile, split) slab costs the launch and its reduce pass, in step times (weight gradients: a
// 256 KB slab written once and read once against a 2.8 us strip step = 0.03).
static int choose_splits_k(long tiles, int max_splits, int slots, long ksteps, double slab_steps = 0.0) {
    if (max_splits < 1) max_splits = 1;
    int best = 1;
    double best_cost = 1e300;
    for (int sp = 1; sp <= max_splits; ++sp) {
        const long blocks = tiles * sp;
        const long rounds = (blocks + slots - 1) / slots;
        const double cost = (double)rounds * ((double)((ksteps + sp - 1) / sp) + 4.0) + slab_steps * (double)blocks;
        if (cost < best_cost * 0.995) { best_cost = cost; best = sp; }
    }
    return best;
}

// XCD-aware work remap (bijective): hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so
// ids b and b+8 share an L2.  Map id -> work index such that each XCD gets a CONTIGUOUS range of work items;
// callers order work items so that neighbours share operands (same pixels, different channel tiles).
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int xcd_remap(int id, int total) {
    const int xcd = id & 7, slot = id >> 3;
    const int qd = total >> 3, rm = total & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
}

// gv_pre / rs_pre / bias_pre: operands the caller fetched ahead of its stores (else they are loaded here).  On gfx9 vmcnt
// counts stores as well as loads, so a load issued after a store waits for that store's write acknowledgement.
__device__ __forceinline__ f32x4 epilogue4(const IGemmParams& p, f32x4 o, size_t pix, int nb, const f32x4* gv_pre = nullptr,
                                           const float* rs_pre = nullptr, const f32x4* bias_pre = nullptr) {
    const float rs = rs_pre ? *rs_pre : (p.rowscale ? p.rowscale[pix] : 1.f);
    if (bias_pre) o += *bias_pre;
    else if (p.bias) o += *reinterpret_cast<const f32x4*>(p.bias + nb);
    o *= rs;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = apply_act(o[e], p.act, p.slope);
    if (p.gate) {
        const f32x4 gv = gv_pre ? *gv_pre : *reinterpret_cast<const f32x4*>(p.gate + pix * p.N + nb);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] *= gv[e] > 0.f ? 1.f : (p.gate_act == TG_ACT_LEAKY ? p.gate_slope : 0.f);
    }
    if (p.accumulate) o += *reinterpret_cast<const f32x4*>(p.dst + pix * p.N + nb);
    return o;
}

__device__ __forceinline__ size_t dst_pixel(const IGemmParams& p, int m) {
    if (p.ds == 1 && p.OH == p.DH && p.OW == p.DW) return (size_t)m;
    int ox = m % p.OW;
    int t = m / p.OW;
    int oy = t % p.OH;
    int b = t / p.OH;
    return ((size_t)b * p.DH + (oy * p.ds + p.dy0)) * p.DW + (ox * p.ds + p.dx0);
}

template <int WAVES_M, int WAVES_N, int WM, int WN, bool SCALAR, bool BF16>
__device__ __forceinline__ void igemm_body(const IGemmParams& p, const int bz) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, LDK = 36;
    constexpr int A_LOADS = BM / 32, B_LOADS = BN / 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                 // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;  // [2][BN][LDK]
    constexpr int LDH = 40;           // bf16 variant: [2][BM][LDH] + [2][BN][LDH] bf16 (see pgemm_kernel)
    __bf16* Ah = reinterpret_cast<__bf16*>(smem);
    __bf16* Bh = Ah + 2 * BM * LDH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= p.M) return;          // merged multi-class launch: the grid is sized for the largest class
    const int kc = tid & 7, r0 = tid >> 3;

    int rb[A_LOADS], ry[A_LOADS], rx[A_LOADS];
    bool rv[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
        int m = m0 + r0 + 32 * i;
        rv[i] = m < p.M;
        int mm = rv[i] ? m : 0;
        int ox = mm % p.OW;
        int t = mm / p.OW;
        int oy = t % p.OH;
        rb[i] = t / p.OH;
        ry[i] = oy * p.ss + p.sy0;
        rx[i] = ox * p.ss + p.sx0;
    }

    const int t_begin = bz * p.steps_per_split;
    const int t_end = min(p.T, t_begin + p.steps_per_split);

    f32x4 ra[A_LOADS], rw[B_LOADS];
    // vector path: the source-mask value of a gathered quad is only FETCHED with it (rmk) and multiplied in when the step is
    // stored to LDS.  With `if (ok) { v = load; v *= mask[pix]; }` every gather waited for its own two requests inside its
    // branch before the next one could issue: a masked K step's loads went out one round trip after the other (enc4 forward
    // 120 -> 117 us, its weight gradient 143 -> 131 us).  Out-of-range taps still issue no load at all: making the loads
    // unconditional at clamped addresses helps the masked layers as much and costs the small-image dgrads 5-10 %.
    float rmk[A_LOADS];

    // loader state for the NEXT K step (vector path): tap (l_ty, l_tx) and channel chunk l_ch, advanced
    // incrementally -- no integer division in the loop
    int l_ty = 0, l_tx = 0, l_ch = 0;
    if constexpr (!SCALAR) {
        const int tap0 = t_begin / p.nchunks;
        l_ch = t_begin - tap0 * p.nchunks;
        l_ty = tap0 / p.TW;
        l_tx = tap0 - l_ty * p.TW;
    }
    int g_c0 = 0, g_dyy = 0, g_dxx = 0, g_widx = 0;
    bool g_cv = false;
    // latch the addressing of the step being loaded, then advance the state
    auto gbegin = [&]() {
        g_c0 = l_ch * 32 + 4 * kc;
        g_dyy = l_ty * p.tstep;
        g_dxx = l_tx * p.tstep;
        g_widx = ((p.ky0 + l_ty * p.kstep) * p.KW + (p.kx0 + l_tx * p.kstep)) * p.C + g_c0;
        g_cv = g_c0 < p.C;
        if (++l_ch == p.nchunks) {
            l_ch = 0;
            if (++l_tx == p.TW) { l_tx = 0; ++l_ty; }
        }
    };
    // quarter `part` (0..3) of the step's global loads: A rows i = part (mod 4), B rows j = part (mod 4)
    auto gpart = [&](int part) {
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            if ((i & 3) != part) continue;
            int iy = ry[i] + g_dyy, ix = rx[i] + g_dxx;
            const bool ok = rv[i] && g_cv && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            float mk = 1.f;
            if (ok) {
                size_t pix = ((size_t)rb[i] * p.IH + iy) * p.IW + ix;
                v = *reinterpret_cast<const f32x4*>(p.src + pix * p.C + g_c0);
                if (p.amask) mk = p.amask[pix];
            }
            ra[i] = v;
            rmk[i] = mk;
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            if ((j & 3) != part) continue;
            int n = n0 + r0 + 32 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < p.N && g_cv) v = *reinterpret_cast<const f32x4*>(p.wmat + (size_t)n * p.Kfull + g_widx);
            rw[j] = v;
        }
    };
    // (vector path) the source mask of the step about to be stored
    auto gfinish = [&]() {
        if constexpr (!SCALAR) {
            if (p.amask) {
#pragma unroll
                for (int i = 0; i < A_LOADS; ++i) ra[i] *= rmk[i];
            }
        }
    };
    auto gload_scalar = [&](int t) {
        // any C (1, 3, ...): K index decoded per element
        int tapv[4], cval[4], dyv[4], dxv[4], wv[4];
        bool kv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int k = t * 32 + 4 * kc + e;
            kv[e] = k < p.Ktot;
            int kk = kv[e] ? k : 0;
            tapv[e] = kk / p.C;
            cval[e] = kk - tapv[e] * p.C;
            int ty = tapv[e] / p.TW, tx = tapv[e] - ty * p.TW;
            dyv[e] = ty * p.tstep;
            dxv[e] = tx * p.tstep;
            wv[e] = ((p.ky0 + ty * p.kstep) * p.KW + (p.kx0 + tx * p.kstep)) * p.C + cval[e];
        }
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int iy = ry[i] + dyv[e], ix = rx[i] + dxv[e];
                bool ok = rv[i] && kv[e] && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                if (ok) {
                    size_t pix = ((size_t)rb[i] * p.IH + iy) * p.IW + ix;
                    float sv = p.src[pix * p.C + cval[e]];
                    if (p.amask) sv *= p.amask[pix];
                    v[e] = sv;
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            int n = n0 + r0 + 32 * j;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n < p.N && kv[e]) v[e] = p.wmat[(size_t)n * p.Kfull + wv[e]];
            rw[j] = v;
        }
    };
    auto gload = [&](int t) {
        if constexpr (!SCALAR) {
            gbegin();
#pragma unroll
            for (int part = 0; part < 4; ++part) gpart(part);
        } else {
            gload_scalar(t);
        }
    };
    auto sstore = [&](int buf) {
        gfinish();
        if constexpr (BF16) {
            __bf16* Ab = Ah + buf * BM * LDH;
            __bf16* Bb = Bh + buf * BN * LDH;
#pragma unroll
            for (int i = 0; i < A_LOADS; ++i)
                *reinterpret_cast<bf16x4*>(Ab + (r0 + 32 * i) * LDH + 4 * kc) = __builtin_convertvector(ra[i], bf16x4);
#pragma unroll
            for (int j = 0; j < B_LOADS; ++j)
                *reinterpret_cast<bf16x4*>(Bb + (r0 + 32 * j) * LDH + 4 * kc) = __builtin_convertvector(rw[j], bf16x4);
            return;
        }
        float* Ab = As + buf * BM * LDK;
        float* Bb = Bs + buf * BN * LDK;
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i)
            *reinterpret_cast<f32x4*>(Ab + (r0 + 32 * i) * LDK + 4 * kc) = ra[i];
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j)
            *reinterpret_cast<f32x4*>(Bb + (r0 + 32 * j) * LDK + 4 * kc) = rw[j];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int arow = wm * WM * 32 + (lane & 31);
    const int brow = wn * WN * 32 + (lane & 31);
    const int ko = 4 * (lane >> 5);

    if (t_begin < t_end) {
        gload(t_begin);
        sstore(0);
        __syncthreads();
        for (int t = t_begin; t < t_end; ++t) {
            const int cur = (t - t_begin) & 1;
            const bool more = (t + 1) < t_end;
            if constexpr (SCALAR) {
                if (more) gload_scalar(t + 1);
            } else {
                if (more) gbegin();
            }
            if constexpr (BF16) {
                if constexpr (!SCALAR) {
                    if (more) {
#pragma unroll
                        for (int part = 0; part < 4; ++part) gpart(part);
                    }
                }
                const __bf16* Ab = Ah + cur * BM * LDH;
                const __bf16* Bb = Bh + cur * BN * LDH;
                const int kh8 = 8 * (lane >> 5);
#pragma unroll
                for (int gk = 0; gk < 2; ++gk) {
                    bf16x8 ah[WM], bh[WN];
#pragma unroll
                    for (int i = 0; i < WM; ++i) ah[i] = *reinterpret_cast<const bf16x8*>(Ab + (arow + 32 * i) * LDH + 16 * gk + kh8);
#pragma unroll
                    for (int j = 0; j < WN; ++j) bh[j] = *reinterpret_cast<const bf16x8*>(Bb + (brow + 32 * j) * LDH + 16 * gk + kh8);
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
            } else {
            const float* Ab = As + cur * BM * LDK;
            const float* Bb = Bs + cur * BN * LDK;
            // operand fragments are double-buffered in registers: the LDS reads of k-group g+1 are issued before the
            // MFMAs of k-group g, so their latency hides behind 16 MFMAs instead of stalling every group
            f32x4 a[2][WM], b[2][WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(Ab + (arow + 32 * i) * LDK + ko);
#pragma unroll
            for (int j = 0; j < WN; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(Bb + (brow + 32 * j) * LDK + ko);
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                // a quarter of the next step's global loads rides behind each MFMA group
                if constexpr (!SCALAR) {
                    if (more) gpart(kg);
                }
                if (kg < 3) {
#pragma unroll
                    for (int i = 0; i < WM; ++i)
                        a[(kg + 1) & 1][i] = *reinterpret_cast<const f32x4*>(Ab + (arow + 32 * i) * LDK + (kg + 1) * 8 + ko);
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        b[(kg + 1) & 1][j] = *reinterpret_cast<const f32x4*>(Bb + (brow + 32 * j) * LDK + (kg + 1) * 8 + ko);
                }
                __builtin_amdgcn_sched_barrier(0);     // keep the prefetch (LDS + global) ABOVE this group's MFMAs
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg & 1][i][e], b[kg & 1][j][e], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            if (more) sstore(cur ^ 1);
            __syncthreads();
        }
    }

    if ((p.N & 3) == 0) {     // wide stores (see tile_rows4); all waves are past the K loop's last barrier
        float* scratch = smem + wave * (32 * 36);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                tile_rows4(scratch, acc[i][j], lane, [&](int rr, int cc, f32x4 v4) {
                    const int m = m0 + (wm * WM + i) * 32 + rr;
                    const int nb = n0 + (wn * WN + j) * 32 + cc;
                    if (m >= p.M || nb >= p.N) return;
                    if (p.splits > 1) {
                        *reinterpret_cast<f32x4*>(p.ws + ((size_t)bz * p.M + m) * p.N + nb) = v4;
                    } else {
                        const size_t pix = dst_pixel(p, m);
                        *reinterpret_cast<f32x4*>(p.dst + pix * p.N + nb) = epilogue4(p, v4, pix, nb);
                    }
                });
        return;
    }
    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int m = m0 + (wm * WM + i) * 32 + row;
            if (m >= p.M) continue;
            if (p.splits > 1) {
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int n = n0 + (wn * WN + j) * 32 + (lane & 31);
                    if (n < p.N) p.ws[((size_t)bz * p.M + m) * p.N + n] = acc[i][j][r];
                }
            } else {
                const size_t pix = dst_pixel(p, m);
                const float rs = p.rowscale ? p.rowscale[pix] : 1.f;
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int n = n0 + (wn * WN + j) * 32 + (lane & 31);
                    if (n < p.N) {
                        float v = acc[i][j][r];
                        if (p.bias) v += p.bias[n];
                        v = apply_act(v * rs, p.act, p.slope);
                        if (p.gate) v *= gate_factor(p, pix * p.N + n);
                        float* d = p.dst + pix * p.N + n;
                        if (p.accumulate) v += *d;
                        *d = v;
                    }
                }
            }
        }
    }
}

template <int WAVES_M, int WAVES_N, int WM, int WN, bool SCALAR, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IGemmParams p) {
    igemm_body<WAVES_M, WAVES_N, WM, WN, SCALAR, BF16>(p, blockIdx.z);
}
// Up to 4 independent problems of one tile configuration in ONE launch (IGemmMulti; the parity classes of a stride-2 dgrad on a
// small grid: four 17-us launches + four split-K epilogues per layer were pure launch latency): blockIdx.z = class * splits + split.
template <int WAVES_M, int WAVES_N, int WM, int WN, bool BF16>
__global__ __launch_bounds__(256, 2) void igemm_multi_kernel(const IGemmMulti pm) {
    int cls = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) cls += (int)blockIdx.z >= pm.zbeg[i] ? 1 : 0;
    igemm_body<WAVES_M, WAVES_N, WM, WN, false, BF16>(pm.c[cls], blockIdx.z - pm.zbeg[cls]);
}

// split-K second pass: fixed-order sum over the slabs + the same epilogue.
__device__ __forceinline__ void splitk_epilogue_body(const IGemmParams& p) {
    const size_t total = (size_t)p.M * p.N;
    if ((p.N & 3) == 0) {
        const size_t t4 = total >> 2;
        for (size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x; i4 < t4; i4 += (size_t)gridDim.x * 256) {
            const size_t idx = i4 * 4;
            const int m = (int)(idx / p.N), n = (int)(idx - (size_t)m * p.N);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            // eight slabs in flight, summed in slab order (small-batch launches have up to 58 slabs and a few thousand
            // threads: one dependent load after the other made this pass 10-19 us)
            const float* wp = p.ws + idx;
            int z = 0;
            for (; z + 8 <= p.splits; z += 8) {
                f32x4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(wp + (size_t)(z + u) * total);
#pragma unroll
                for (int u = 0; u < 8; ++u) v += t[u];
            }
            if (z < p.splits) {
                // the last (or only) group: up to seven slabs, again all in flight (clamped index + select: a rolled tail waits
                // for one slab after the other, and the per-class split plans of the merged dgrads have 2 ... 7 slabs)
                f32x4 t[7];
#pragma unroll
                for (int u = 0; u < 7; ++u) t[u] = *reinterpret_cast<const f32x4*>(wp + (size_t)(z + u < p.splits ? z + u : z) * total);
#pragma unroll
                for (int u = 0; u < 7; ++u)
                    if (z + u < p.splits) v += t[u];
            }
            const size_t pix = dst_pixel(p, m);
            *reinterpret_cast<f32x4*>(p.dst + pix * p.N + n) = epilogue4(p, v, pix, n);
        }
        return;
    }
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / p.N), n = (int)(idx - (size_t)m * p.N);
        float v = 0.f;
        for (int z = 0; z < p.splits; ++z) v += p.ws[(size_t)z * total + idx];
        const size_t pix = dst_pixel(p, m);
        if (p.bias) v += p.bias[n];
        if (p.rowscale) v *= p.rowscale[pix];
        v = apply_act(v, p.act, p.slope);
        if (p.gate) v *= gate_factor(p, pix * p.N + n);
        float* d = p.dst + pix * p.N + n;
        if (p.accumulate) v += *d;
        *d = v;
    }
}

__global__ __launch_bounds__(256) void igemm_splitk_epilogue(const IGemmParams p) { splitk_epilogue_body(p); }
// (a class left at one split has written dst, epilogue applied, from the main kernel: nothing to do for it here)
__global__ __launch_bounds__(256) void igemm_splitk_epilogue_multi(const IGemmMulti pm) {
    if (pm.c[blockIdx.y].splits > 1) splitk_epilogue_body(pm.c[blockIdx.y]);
}

template <int WAVES_M, int WAVES_N, int WM, int WN, bool SCALAR, bool BF16 = false>
static int launch_igemm_cfg(const IGemmParams& p, hipStream_t s) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr size_t lds = (size_t)2 * (BM + BN) * (BF16 ? 40 * 2 : 36 * sizeof(float));
    static LdsOptIn opt;
    auto kern = igemm_kernel<WAVES_M, WAVES_N, WM, WN, SCALAR, BF16>;
    if (int rc = lds_opt_in(opt, reinterpret_cast<const void*>(kern), lds, "igemm")) return rc;
    dim3 grid(cdiv(p.M, BM), cdiv(p.N, BN), p.splits);
    {
        // algorithmic bytes: source pixels touched once + row scale/mask + weights + output
        const double by = 4.0 * ((double)p.B * p.IH * p.IW * p.C + (double)p.M + (double)p.N * p.Ktot + (double)p.M * p.N +
                                 (p.amask ? (double)p.B * p.IH * p.IW : 0.0));
        ProfScope ps(s, BF16 ? 3 : 0, 2.0 * p.M * (double)p.N * p.Ktot, by, p.M, p.N, p.Ktot, p.C, p.splits,
                     (BF16 ? 3000 : 0) + BN + (SCALAR ? 1 : 0));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
    }
    TG_CHECK_LAUNCH("igemm_kernel");
    if (p.splits > 1) {
        hipLaunchKernelGGL(igemm_splitk_epilogue, dim3(ew_grid((int64_t)p.M * p.N, 256)), dim3(256), 0, s, p);
        TG_CHECK_LAUNCH("igemm_splitk_epilogue");
    }
    return TG_OK;
}


// ------------------------------------------------------------------------------------------------
// v2: patch-staged implicit GEMM for stride-1 gathers (every stride-1 forward conv and EVERY dgrad,
// whose parity classes are stride-1 over the dy grid).  The workgroup owns a TH x TW block of output
// pixels of one image; for each 32-channel chunk the (TH+taps-1) x (TW+taps-1) input patch is staged
// into LDS ONCE (mask multiply and bounds handling happen there, once per pixel) and all taps read their
// A fragments straight from it at a shifted base -- 9x fewer A-side global loads and index math for a
// 3x3 conv than gathering per tap.  Only the weight tile streams per (chunk, tap), double-buffered.
// ------------------------------------------------------------------------------------------------
// One output-grid "class": a whole stride-1 problem, or one parity class of a strided dgrad.  Up to 4 classes
// (stride 2) share one launch so that the grid fills the chip without split-K.
struct ClassGeom {
    int OH, OW, dy0, dx0;                // output grid and its placement in dst
    int ky0, kx0, TH, TW, sy0, sx0;      // taps of this class
    int tiles_x, tiles_y;                // output tiles per image
    int PH, PW;                          // patch size in source pixels
    int sy_min, sx_min;                  // min over taps of (sy0 + ty*tstep), (sx0 + tx*tstep)
    int work_begin;                      // first flat work item (patch tile x N tile) of this class
};
struct PatchGeom {
    int ncls, total_work, chunks_per_split;
    ClassGeom c[4];
};

// BF16 = true: operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) when they are staged into LDS and multiplied on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (16x the fp32 MFMA rate); activations, weights, masks, bias and
// the whole epilogue stay fp32 (BASELINE config 3: bf16 compute, fp32 masters).  LDS rows are 32 bf16 + 8 pad (80 B):
// the 16-byte fragment fetch (8 consecutive k per lane) is bank-conflict free.
template <int TH_, int TW_, int WAVES_M, int WAVES_N, int WM, int WN, int MAXPL, bool BF16>
__global__ __launch_bounds__(256, 2) void pgemm_kernel(const IGemmParams p, const PatchGeom q) {
    constexpr int BM = TH_ * TW_, BN = WAVES_N * WN * 32, LDK = 36, LDH = 40;
    constexpr int B_LOADS = BN / 32;
    static_assert(BM == WAVES_M * WM * 32 && WAVES_M * WAVES_N == 4, "tile/wave mismatch");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;                    // [2][BN][LDK]
    float* Ps = smem + 2 * BN * LDK;     // [PH*PW][LDK]
    __bf16* Bh = reinterpret_cast<__bf16*>(smem);     // bf16 images: [2][BN][LDH] then [PH*PW][LDH]
    __bf16* Ph = Bh + 2 * BN * LDH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int kc = tid & 7, r0 = tid >> 3;
    // flat (class, patch tile, N tile) index, XCD-remapped so the N tiles of one patch share an L2
    const int nty = cdiv_dev(p.N, BN);
    int work = xcd_remap(blockIdx.x, q.total_work);
    int ci = 0;
    while (ci + 1 < q.ncls && work >= q.c[ci + 1].work_begin) ++ci;
    const ClassGeom g = q.c[ci];
    work -= g.work_begin;
    const int n0 = (work % nty) * BN;
    int tile = work / nty;
    const int txi = tile % g.tiles_x;
    tile /= g.tiles_x;
    const int tyi = tile % g.tiles_y;
    const int b = tile / g.tiles_y;
    const int oy0 = tyi * TH_, ox0 = txi * TW_;
    const int py0 = oy0 * p.ss + g.sy_min, px0 = ox0 * p.ss + g.sx_min;
    const int ppix = g.PH * g.PW;

    // patch slots owned by this thread: fixed for the whole K loop
    uint32_t poff[MAXPL];
    float pmul[MAXPL];
#pragma unroll
    for (int i = 0; i < MAXPL; ++i) {
        const int idx = tid + 256 * i;
        const int pp = idx >> 3;
        poff[i] = 0;
        pmul[i] = 0.f;
        if (pp < ppix) {
            const int py = pp / g.PW, px = pp - py * g.PW;
            const int iy = py0 + py, ix = px0 + px;
            if (iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) {
                const uint32_t pix = ((uint32_t)b * p.IH + iy) * p.IW + ix;
                poff[i] = pix * (uint32_t)p.C + 4 * kc;
                pmul[i] = p.amask ? p.amask[pix] : 1.f;
            }
        }
    }
    uint32_t boff[B_LOADS];
    bool bok[B_LOADS];
#pragma unroll
    for (int j = 0; j < B_LOADS; ++j) {
        const int n = n0 + r0 + 32 * j;
        bok[j] = n < p.N;
        boff[j] = (uint32_t)(bok[j] ? n : 0) * (uint32_t)p.Kfull + 4 * kc;
    }
    // A-fragment bases: row r of the M tile is output pixel (r / TW, r % TW)
    int abase[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int r = (wm * WM + i) * 32 + (lane & 31);
        abase[i] = ((r / TW_) * p.ss * g.PW + (r % TW_) * p.ss) * (BF16 ? LDH : LDK) + (BF16 ? 8 : 4) * (lane >> 5);
    }
    const int brow = wn * WN * 32 + (lane & 31);
    const int ko = 4 * (lane >> 5);
    const int ntaps = g.TH * g.TW;
    const int c_begin = blockIdx.z * q.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + q.chunks_per_split);

    // Patch staging registers.  Small patches (<= 8 slots/thread) are prefetched during the last tap of the
    // previous chunk; the 16x16 tile (up to 12 slots) would push the kernel past 256 VGPRs, so it reloads its
    // patch in two register batches behind the chunk-boundary barrier instead.
    constexpr bool PREFETCH = MAXPL <= 8;
    constexpr int PB = PREFETCH ? MAXPL : (MAXPL + 1) / 2;      // slots per register batch
    f32x4 rp[PB], rw[B_LOADS];
    auto pload = [&](int c, int batch) {
        const bool cv = c * 32 + 4 * kc < p.C;
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int sl = batch * PB + i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (sl < MAXPL && pmul[sl] != 0.f && cv) v = *reinterpret_cast<const f32x4*>(p.src + poff[sl] + c * 32) * pmul[sl];
            rp[i] = v;
        }
    };
    auto pstore = [&](int batch) {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
            const int sl = batch * PB + i;
            const int idx = tid + 256 * sl;
            if (sl < MAXPL && (idx >> 3) < ppix) {
                if constexpr (BF16) *reinterpret_cast<bf16x4*>(Ph + (idx >> 3) * LDH + 4 * kc) = __builtin_convertvector(rp[i], bf16x4);
                else *reinterpret_cast<f32x4*>(Ps + (idx >> 3) * LDK + 4 * kc) = rp[i];
            }
        }
    };
    constexpr int NBATCH = PREFETCH ? 1 : 2;
    auto wload = [&](int c, int tap) {
        const int ty = tap / g.TW, tx = tap - ty * g.TW;
        const uint32_t widx = (uint32_t)(((g.ky0 + ty * p.kstep) * p.KW + (g.kx0 + tx * p.kstep)) * p.C + c * 32);
        const bool cv = c * 32 + 4 * kc < p.C;
#pragma unroll
        for (int j = 0; j < B_LOADS; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (bok[j] && cv) v = *reinterpret_cast<const f32x4*>(p.wmat + boff[j] + widx);
            rw[j] = v;
        }
    };
    auto wstore = [&](int buf) {
        if constexpr (BF16) {
            __bf16* Bb = Bh + buf * BN * LDH;
#pragma unroll
            for (int j = 0; j < B_LOADS; ++j)
                *reinterpret_cast<bf16x4*>(Bb + (r0 + 32 * j) * LDH + 4 * kc) = __builtin_convertvector(rw[j], bf16x4);
        } else {
            float* Bb = Bs + buf * BN * LDK;
#pragma unroll
            for (int j = 0; j < B_LOADS; ++j) *reinterpret_cast<f32x4*>(Bb + (r0 + 32 * j) * LDK + 4 * kc) = rw[j];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (c_begin < c_end && ntaps > 0) {
        wload(c_begin, 0);
#pragma unroll
        for (int bt = 0; bt < NBATCH; ++bt) {
            pload(c_begin, bt);
            pstore(bt);
        }
        wstore(0);
        __syncthreads();
        int cur = 0;
        for (int c = c_begin; c < c_end; ++c) {
            for (int tap = 0; tap < ntaps; ++tap) {
                const bool last_tap = tap + 1 == ntaps;
                const bool next_chunk = last_tap && (c + 1 < c_end);
                const bool more = !last_tap || next_chunk;
                if (more) wload(last_tap ? c + 1 : c, last_tap ? 0 : tap + 1);
                if (next_chunk) pload(c + 1, 0);      // first register batch of the next patch rides under this tap's MFMAs
                const int ty = tap / g.TW, tx = tap - ty * g.TW;
                const int toff = ((g.sy0 + ty * p.tstep - g.sy_min) * g.PW + (g.sx0 + tx * p.tstep - g.sx_min)) * (BF16 ? LDH : LDK);
                if constexpr (BF16) {
                    const __bf16* Bb = Bh + cur * BN * LDH;
                    const int kh8 = 8 * (lane >> 5);
                    bf16x8 ah[2][WM], bh[2][WN];
#pragma unroll
                    for (int i = 0; i < WM; ++i) ah[0][i] = *reinterpret_cast<const bf16x8*>(Ph + abase[i] + toff);
#pragma unroll
                    for (int j = 0; j < WN; ++j) bh[0][j] = *reinterpret_cast<const bf16x8*>(Bb + (brow + 32 * j) * LDH + kh8);
#pragma unroll
                    for (int i = 0; i < WM; ++i) ah[1][i] = *reinterpret_cast<const bf16x8*>(Ph + abase[i] + toff + 16);
#pragma unroll
                    for (int j = 0; j < WN; ++j) bh[1][j] = *reinterpret_cast<const bf16x8*>(Bb + (brow + 32 * j) * LDH + 16 + kh8);
#pragma unroll
                    for (int gk = 0; gk < 2; ++gk)
#pragma unroll
                        for (int i = 0; i < WM; ++i)
#pragma unroll
                            for (int j = 0; j < WN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[gk][i], bh[gk][j], acc[i][j], 0, 0, 0);
                } else {
                const float* Bb = Bs + cur * BN * LDK;
                f32x4 a[2][WM], bb[2][WN];       // register double-buffered fragments (see igemm_kernel)
#pragma unroll
                for (int i = 0; i < WM; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(Ps + abase[i] + toff);
#pragma unroll
                for (int j = 0; j < WN; ++j) bb[0][j] = *reinterpret_cast<const f32x4*>(Bb + (brow + 32 * j) * LDK + ko);
#pragma unroll
                for (int kg = 0; kg < 4; ++kg) {
                    if (kg < 3) {
#pragma unroll
                        for (int i = 0; i < WM; ++i)
                            a[(kg + 1) & 1][i] = *reinterpret_cast<const f32x4*>(Ps + abase[i] + toff + (kg + 1) * 8);
#pragma unroll
                        for (int j = 0; j < WN; ++j)
                            bb[(kg + 1) & 1][j] = *reinterpret_cast<const f32x4*>(Bb + (brow + 32 * j) * LDK + (kg + 1) * 8 + ko);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < WM; ++i)
#pragma unroll
                            for (int j = 0; j < WN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kg & 1][i][e], bb[kg & 1][j][e], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
                if (next_chunk) {
                    __syncthreads();        // every wave is done with the current patch
                    pstore(0);
#pragma unroll
                    for (int bt = 1; bt < NBATCH; ++bt) {     // 16x16 tile: second batch behind the barrier
                        pload(c + 1, bt);
                        pstore(bt);
                    }
                }
                if (more) wstore(cur ^ 1);
                __syncthreads();
                cur ^= 1;
            }
        }
    }

    // epilogue: row r -> output pixel (oy0 + r / TW, ox0 + r % TW); wide stores via tile_rows4 when N % 4 == 0
    if ((p.N & 3) == 0) {
        float* scratch = smem + wave * (32 * 36);     // the K loop is over and every wave has passed its last barrier
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                tile_rows4(scratch, acc[i][j], lane, [&](int rr, int cc, f32x4 v4) {
                    const int row = (wm * WM + i) * 32 + rr;
                    const int nb = n0 + (wn * WN + j) * 32 + cc;
                    const int oy = oy0 + row / TW_, ox = ox0 + row % TW_;
                    if (oy >= g.OH || ox >= g.OW || nb >= p.N) return;
                    if (p.splits > 1) {
                        const size_t m = ((size_t)b * g.OH + oy) * g.OW + ox;
                        *reinterpret_cast<f32x4*>(p.ws + ((size_t)blockIdx.z * p.M + m) * p.N + nb) = v4;
                    } else {
                        const size_t pix = ((size_t)b * p.DH + (oy * p.ds + g.dy0)) * p.DW + (ox * p.ds + g.dx0);
                        *reinterpret_cast<f32x4*>(p.dst + pix * p.N + nb) = epilogue4(p, v4, pix, nb);
                    }
                });
        return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * WM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int oy = oy0 + row / TW_, ox = ox0 + row % TW_;
            if (oy >= g.OH || ox >= g.OW) continue;
            if (p.splits > 1) {
                const size_t m = ((size_t)b * g.OH + oy) * g.OW + ox;
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int n = n0 + (wn * WN + j) * 32 + (lane & 31);
                    if (n < p.N) p.ws[((size_t)blockIdx.z * p.M + m) * p.N + n] = acc[i][j][r];
                }
            } else {
                const size_t pix = ((size_t)b * p.DH + (oy * p.ds + g.dy0)) * p.DW + (ox * p.ds + g.dx0);
                const float rs = p.rowscale ? p.rowscale[pix] : 1.f;
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int n = n0 + (wn * WN + j) * 32 + (lane & 31);
                    if (n < p.N) {
                        float v = acc[i][j][r];
                        if (p.bias) v += p.bias[n];
                        v = apply_act(v * rs, p.act, p.slope);
                        if (p.gate) v *= gate_factor(p, pix * p.N + n);
                        float* d = p.dst + pix * p.N + n;
                        if (p.accumulate) v += *d;
                        *d = v;
                    }
                }
            }
        }
    }
}

template <int TH_, int TW_, int WAVES_M, int WAVES_N, int WM, int WN, int MAXPL, bool BF16>
static int launch_pgemm_cfg(const IGemmParams& p, const PatchGeom& q, double flops, double bytes, int Mtot, hipStream_t s) {
    constexpr int BN = WAVES_N * WN * 32;
    int ppix = 0;
    for (int i = 0; i < q.ncls; ++i) ppix = q.c[i].PH * q.c[i].PW > ppix ? q.c[i].PH * q.c[i].PW : ppix;
    const size_t lds = ((size_t)2 * BN + (size_t)ppix) * (BF16 ? 40 * 2 : 36 * sizeof(float));
    static LdsOptIn opt;
    auto kern = pgemm_kernel<TH_, TW_, WAVES_M, WAVES_N, WM, WN, MAXPL, BF16>;
    constexpr size_t LDS_MAX = 96 * 1024;       // the opt-in is a ceiling (largest patch: 19 x 19 pixels), not an allocation
    if (lds > LDS_MAX) { tg_set_error("pgemm: %zu bytes of LDS requested", lds); return TG_ERR_ARG; }
    if (int rc = lds_opt_in(opt, reinterpret_cast<const void*>(kern), LDS_MAX, "pgemm")) return rc;
    dim3 grid(q.total_work, 1, p.splits);
    {
        ProfScope ps(s, BF16 ? 3 : 0, flops, bytes, Mtot, p.N, p.Ktot, p.C, p.splits, (BF16 ? 3000 : 1000) + BN);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p, q);
    }
    TG_CHECK_LAUNCH("pgemm_kernel");
    if (p.splits > 1) {
        hipLaunchKernelGGL(igemm_splitk_epilogue, dim3(ew_grid((int64_t)p.M * p.N, 256)), dim3(256), 0, s, p);
        TG_CHECK_LAUNCH("igemm_splitk_epilogue");
    }
    return TG_OK;
}

static bool pgemm_class_ok(const IGemmParams& p) {
    if (p.ss != 1 || (p.C % 32) != 0 || p.TH < 1 || p.TW < 1 || p.TH > 4 || p.TW > 4) return false;
    if (p.OW < 16 || p.OH < 8) return false;
    const bool n64 = !(p.N >= 128 && p.N % 128 == 0);
    if (n64 && (p.N < 48 || p.OH < 16)) return false;       // tiny N: dedicated kernels / v1
    return true;
}

// Patch kernel for `ncls` classes (1 = ordinary conv / stride-1 dgrad, 4 = the parity classes of a stride-2 dgrad
// merged into one launch).  Returns true (and launches) when it applies.
static bool try_pgemm(IGemmParams* ps, int ncls, size_t ws_floats_avail, hipStream_t s, int* rc) {
    if (getenv("TG_NO_PGEMM") || ncls < 1 || ncls > 4) return false;
    for (int i = 0; i < ncls; ++i)
        if (!pgemm_class_ok(ps[i])) return false;
    IGemmParams& p = ps[0];
    const bool n64 = !(p.N >= 128 && p.N % 128 == 0);
    const int th = n64 ? 16 : 8, tw = 16, bn = n64 ? 64 : 128;
    const int nty = cdiv(p.N, bn);
    PatchGeom q = {};
    q.ncls = ncls;
    int work = 0, Mtot = 0;
    double flops = 0, bytes = 0;
    for (int i = 0; i < ncls; ++i) {
        const IGemmParams& pc = ps[i];
        ClassGeom& c = q.c[i];
        c.OH = pc.OH; c.OW = pc.OW; c.dy0 = pc.dy0; c.dx0 = pc.dx0;
        c.ky0 = pc.ky0; c.kx0 = pc.kx0; c.TH = pc.TH; c.TW = pc.TW; c.sy0 = pc.sy0; c.sx0 = pc.sx0;
        c.tiles_x = cdiv(pc.OW, tw);
        c.tiles_y = cdiv(pc.OH, th);
        const int sy_b = pc.sy0 + (pc.TH - 1) * pc.tstep, sx_b = pc.sx0 + (pc.TW - 1) * pc.tstep;
        c.sy_min = pc.sy0 < sy_b ? pc.sy0 : sy_b;
        c.sx_min = pc.sx0 < sx_b ? pc.sx0 : sx_b;
        c.PH = th + pc.TH - 1;
        c.PW = tw + pc.TW - 1;
        c.work_begin = work;
        work += c.tiles_x * c.tiles_y * pc.B * nty;
        Mtot += pc.M;
        const double kt = (double)pc.TH * pc.TW * pc.C;
        flops += 2.0 * pc.M * (double)pc.N * kt;
        bytes += 4.0 * ((double)pc.M * pc.N + (double)pc.N * kt);
    }
    bytes += 4.0 * ((double)p.B * p.IH * p.IW * p.C + (double)Mtot + (p.amask ? (double)p.B * p.IH * p.IW : 0.0));
    q.total_work = work;
    p.Ktot = p.TH * p.TW * p.C;
    p.nchunks = p.C / 32;
    int splits = 1;
    if (ncls == 1) {
        if (p.nchunks >= 4) {
            int smax = p.nchunks / 2 < 32 ? p.nchunks / 2 : 32;
            while (smax > 1 && (size_t)smax * p.M * p.N > ws_floats_avail) --smax;
            splits = choose_splits(work, smax, 512);
        }
    } else {
        // merged launch has no split-K: a grid below one full round of resident workgroups (2 per CU) goes to the
        // gathered-row kernel instead, whose merged launch splits K (enc4 dgrad: 256 patch workgroups ran at 50 TF)
        static const int min_work = getenv("TG_PGEMM_MERGE_MIN") ? atoi(getenv("TG_PGEMM_MERGE_MIN")) : 512;
        if (work < min_work) return false;
    }
    q.chunks_per_split = cdiv(p.nchunks, splits);
    p.splits = cdiv(p.nchunks, q.chunks_per_split);
    if (p.bf16) {
        if (n64) *rc = launch_pgemm_cfg<16, 16, 4, 1, 2, 2, 12, true>(p, q, flops, bytes, Mtot, s);
        else *rc = launch_pgemm_cfg<8, 16, 2, 2, 2, 2, 7, true>(p, q, flops, bytes, Mtot, s);
    } else {
        if (n64) *rc = launch_pgemm_cfg<16, 16, 4, 1, 2, 2, 12, false>(p, q, flops, bytes, Mtot, s);
        else *rc = launch_pgemm_cfg<8, 16, 2, 2, 2, 2, 7, false>(p, q, flops, bytes, Mtot, s);
    }
    return true;
}

#include "wino.inc"
#include "wino44.inc"
#include "wino16.inc"

static int pick_bn(int N) { return N >= 128 && N % 128 == 0 ? 128 : (N > 32 ? 64 : 32); }

// K-split so that small-M layers (enc5-7, dec7, dec6) still fill 256 CUs.
static void plan_splits(IGemmParams& p, size_t ws_floats_avail) {
    const int bn = pick_bn(p.N);
    const long tiles = (long)cdiv(p.M, 128) * cdiv(p.N, bn);
    int splits = 1;
    if (p.T >= 8) {
        int smax = p.T / 4 < 64 ? p.T / 4 : 64;
        while (smax > 1 && (size_t)smax * p.M * p.N > ws_floats_avail) --smax;
        splits = choose_splits(tiles, smax, 512);
    }
    p.steps_per_split = cdiv(p.T > 0 ? p.T : 1, splits);
    p.splits = p.T > 0 ? cdiv(p.T, p.steps_per_split) : 1;
}

static int launch_igemm(IGemmParams& p, hipStream_t s, size_t ws_floats_avail = 0) {
    if (p.M <= 0 || p.N <= 0) return TG_OK;
    if (smallconv_fwd_applies(p)) {          // 1-channel side: HBM-bound dedicated kernels (smallconv.hip)
        p.splits = 1;
        p.Ktot = p.TH * p.TW * p.C;
        const double by = 4.0 * ((double)p.B * p.IH * p.IW * p.C + (double)p.M * p.N + (double)p.N * p.Ktot);
        ProfScope ps(s, 2, 2.0 * p.M * (double)p.N * p.Ktot, by, p.M, p.N, p.Ktot, p.C, 1, 2000);
        return smallconv_fwd_launch(p, s);
    }
    if (wino44_ok(p)) return launch_wino44(p, s);
    if (wino_ok(p)) return launch_wino(p, ws_floats_avail, s);
    if (wino16_ok(p)) return launch_wino16(p, ws_floats_avail, s);
    {
        int rc = TG_OK;
        if (try_pgemm(&p, 1, ws_floats_avail, s, &rc)) return rc;
    }
    const bool scalar = (p.C % 4) != 0;
    p.Ktot = p.TH * p.TW * p.C;
    p.nchunks = cdiv(p.C, 32);
    p.T = scalar ? cdiv(p.Ktot, 32) : p.TH * p.TW * p.nchunks;
    const int bn = pick_bn(p.N);
    if (scalar) {
        if (bn == 128) return launch_igemm_cfg<2, 2, 2, 2, true>(p, s);
        if (bn == 64) return launch_igemm_cfg<2, 2, 2, 1, true>(p, s);
        return launch_igemm_cfg<4, 1, 1, 1, true>(p, s);
    }
    if (p.bf16) {
        if (bn == 128) return launch_igemm_cfg<2, 2, 2, 2, false, true>(p, s);
        if (bn == 64) return launch_igemm_cfg<2, 2, 2, 1, false, true>(p, s);
    }
    if (bn == 128) return launch_igemm_cfg<2, 2, 2, 2, false>(p, s);
    if (bn == 64) return launch_igemm_cfg<2, 2, 2, 1, false>(p, s);
    return launch_igemm_cfg<4, 1, 1, 1, false>(p, s);
}

// The parity classes of a strided dgrad that the patch kernel does not take (small grids: enc4-7) as ONE gathered-row
// launch + ONE split-K epilogue launch.  All classes share the tile configuration and the split count; a class with fewer K
// steps than splits leaves its surplus slabs zero.  Returns false when the classes do not qualify (caller goes class by class).
template <int WAVES_M, int WAVES_N, int WM, int WN, bool BF16>
static int launch_igemm_multi_cfg(const IGemmMulti& pm, int ncls, hipStream_t s) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr size_t lds = (size_t)2 * (BM + BN) * (BF16 ? 40 * 2 : 36 * sizeof(float));
    static LdsOptIn opt;
    auto kern = igemm_multi_kernel<WAVES_M, WAVES_N, WM, WN, BF16>;
    if (int rc = lds_opt_in(opt, reinterpret_cast<const void*>(kern), lds, "igemm multi")) return rc;
    int mmax = 0;
    double flops = 0, by = 0;
    for (int i = 0; i < ncls; ++i) {
        const IGemmParams& p = pm.c[i];
        mmax = p.M > mmax ? p.M : mmax;
        flops += 2.0 * p.M * (double)p.N * p.Ktot;
        by += 4.0 * ((double)p.M + (double)p.N * p.Ktot + (double)p.M * p.N);
    }
    by += 4.0 * (double)pm.c[0].B * pm.c[0].IH * pm.c[0].IW * pm.c[0].C;
    int splits = 1;
    for (int i = 0; i < ncls; ++i) splits = pm.c[i].splits > splits ? pm.c[i].splits : splits;
    dim3 grid(cdiv(mmax, BM), cdiv(pm.c[0].N, BN), pm.zbeg[ncls]);
    {
        ProfScope ps(s, BF16 ? 3 : 0, flops, by, mmax * ncls, pm.c[0].N, pm.c[0].Ktot, pm.c[0].C, splits, (BF16 ? 3000 : 0) + 500 + BN);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, pm);
    }
    TG_CHECK_LAUNCH("igemm_multi_kernel");
    if (splits > 1) {
        hipLaunchKernelGGL(igemm_splitk_epilogue_multi, dim3(ew_grid((int64_t)mmax * pm.c[0].N / 4, 256), ncls), dim3(256), 0, s, pm);
        TG_CHECK_LAUNCH("igemm_splitk_epilogue_multi");
    }
    return TG_OK;
}
static bool try_igemm_multi(IGemmParams* cls, int ncls, float* ws, size_t ws_floats, hipStream_t s, int* rc) {
    static const bool off = getenv("TG_NO_IGEMM_MULTI") != nullptr;
    if (off || ncls < 2 || ncls > 4) return false;
    const int bn = pick_bn(cls[0].N);
    if (bn != 128 && bn != 64) return false;
    long tiles = 0;
    int tmax = 0;
    size_t mn = 0;
    for (int i = 0; i < ncls; ++i) {
        const IGemmParams& p = cls[i];
        if (p.M <= 0 || p.N != cls[0].N || (p.C % 4) != 0 || p.bf16 != cls[0].bf16 || smallconv_fwd_applies(p) || p.T <= 0) return false;
        tiles += (long)cdiv(p.M, 128) * cdiv(p.N, bn);
        tmax = p.T > tmax ? p.T : tmax;
        mn += (size_t)p.M * p.N;
    }
    int splits = 1;
    if (tmax >= 8) {
        int smax = tmax / 4 < 64 ? tmax / 4 : 64;
        while (smax > 1 && (size_t)smax * mn > ws_floats) --smax;
        splits = choose_splits(tiles, smax, 512);
    }
    // Split counts PER CLASS.  The parity classes of a 3x3 stride-2 dgrad walk 4, 2, 2 and 1 taps: with one split count for all, the
    // workgroups of the 1-tap class are done in a quarter of the time of the 4-tap class and their slots idle (enc4's dgrad: 512
    // resident workgroups, launch time = the 4-tap class's K walk).  Choose a common walk LENGTH L instead, s_i = ceil(T_i / L),
    // minimising rounds x (longest walk + ~4 steps of prologue / epilogue).
    int sp[4] = {splits, splits, splits, splits};
    static const bool uniform = getenv("TG_MULTI_UNIFORM_SPLITS") != nullptr;
    if (!uniform && tmax >= 8) {
        long tl[4];
        for (int i = 0; i < ncls; ++i) tl[i] = (long)cdiv(cls[i].M, 128) * cdiv(cls[i].N, bn);
        double best = 1e300;
        for (int L = tmax; L >= 4; --L) {
            long blocks = 0;
            int longest = 0, cand[4];
            size_t need = 0;
            for (int i = 0; i < ncls; ++i) {
                int si = cdiv(cls[i].T, L);
                const int cap = cls[i].T / 4 < 1 ? 1 : (cls[i].T / 4 < 64 ? cls[i].T / 4 : 64);
                if (si > cap) si = cap;
                cand[i] = si;
                blocks += tl[i] * si;
                const int walk = cdiv(cls[i].T, si);
                longest = walk > longest ? walk : longest;
                need += (size_t)si * cls[i].M * cls[i].N;
            }
            if (need > ws_floats) break;
            const long rounds = (blocks + 511) / 512;
            const double cost = (double)rounds * (longest + 4.0) + 0.002 * (double)blocks;
            if (cost < best * 0.995) {
                best = cost;
                for (int i = 0; i < ncls; ++i) sp[i] = cand[i];
            }
        }
    }
    IGemmMulti pm = {};
    size_t off_f = 0;
    for (int i = 0; i < ncls; ++i) {
        IGemmParams& p = cls[i];
        p.steps_per_split = cdiv(p.T, sp[i]);
        p.splits = cdiv(p.T, p.steps_per_split);          // (no empty trailing split)
        p.ws = ws + off_f;
        off_f += (size_t)p.splits * p.M * p.N;
        pm.zbeg[i] = i == 0 ? 0 : pm.zbeg[i - 1] + cls[i - 1].splits;
        pm.c[i] = p;
    }
    pm.zbeg[ncls] = pm.zbeg[ncls - 1] + cls[ncls - 1].splits;
    for (int i = ncls; i < 4; ++i) {
        pm.c[i] = pm.c[0];
        pm.zbeg[i + 1] = pm.zbeg[ncls];
    }
    if (cls[0].bf16) *rc = bn == 128 ? launch_igemm_multi_cfg<2, 2, 2, 2, true>(pm, ncls, s) : launch_igemm_multi_cfg<2, 2, 2, 1, true>(pm, ncls, s);
    else *rc = bn == 128 ? launch_igemm_multi_cfg<2, 2, 2, 2, false>(pm, ncls, s) : launch_igemm_multi_cfg<2, 2, 2, 1, false>(pm, ncls, s);
    return true;
}

static int check_conv(const TgConv* g, const char* who) {
    TG_REQUIRE(g != nullptr, "%s: null geometry", who);
    TG_REQUIRE(g->B > 0 && g->H > 0 && g->W > 0 && g->Cin > 0 && g->Cout > 0, "%s: non-positive dims", who);
    TG_REQUIRE(g->k > 0 && g->stride > 0 && g->pad >= 0, "%s: bad k/stride/pad", who);
    TG_REQUIRE(g->Ho == (g->H + 2 * g->pad - g->k) / g->stride + 1 && g->Wo == (g->W + 2 * g->pad - g->k) / g->stride + 1,
               "%s: Ho/Wo (%d,%d) inconsistent with H,W,k,s,p (%d,%d,%d,%d,%d)", who, g->Ho, g->Wo, g->H, g->W, g->k,
               g->stride, g->pad);
    TG_REQUIRE(g->Ho > 0 && g->Wo > 0, "%s: empty output", who);
    TG_REQUIRE((int64_t)g->B * g->H * g->W * (int64_t)g->Cin < (1ll << 31) &&
                   (int64_t)g->B * g->Ho * g->Wo * (int64_t)g->Cout < (1ll << 31),
               "%s: tensor too large for 32-bit pixel indexing", who);
    return TG_OK;
}
static bool aligned16(const void* ptr) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; }

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// room for the Winograd-transformed weights of a stride-1 3x3 layer (0 when the layer can never take that path)
#include "wino22.inc"
static size_t conv_wino_floats(const TgConv* g) {
    if (wino22_fwd_geom_ok(g) || wino22_dgrad_geom_ok(g)) return wino22_u_floats(g);
    if (g->k != 3 || g->stride != 1 || (g->Cin % 8) != 0 || (g->Cout % 8) != 0) return 0;
    if (g->precision == TG_PREC_F32_WINO4) return align_up(wino44_u_floats(g->Cout, g->Cin), 64);    // room for either transform
    return align_up(wino_u_floats(g->Cout, g->Cin), 64);
}

// ---- 5x5 stride-2 layers on the Winograd kernels ---------------------------------------------------
// A 5x5 / stride 2 / pad 2 convolution is EXACTLY a 3x3 / stride 1 / pad 1 convolution over the space-to-depth input
// x2[b][yy][xx][(dy,dx,c)] = x[b][2yy+dy][2xx+dx][c] with weights w2[co][ty][tx][(dy,dx,c)] = w[co][2ty+dy][2tx+dx][c]
// (taps with 2ty+dy = 5 or 2tx+dx = 5 are zero: 25 of 36 live).  1.44x the algorithmic work, but on the Winograd kernels
// (2.25x fewer multiplies, twice the rate of the strided gather kernels) enc2 / enc3 come out ~35 % faster, forward,
// dgrad and wgrad alike.  The rearrangements are one bandwidth-bound pass each; the mask multiply of the partial conv
// rides in the space-to-depth pass, so the Winograd kernels see a pre-masked input.
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ x, const float* __restrict__ mask, float* __restrict__ x2,
                                                  int B, int H, int W, int C) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)B * H * W * C4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        int64_t pix = idx / C4;
        const int xw = (int)(pix % W);
        int64_t t = pix / W;
        const int yh = (int)(t % H);
        const int b = (int)(t / H);
        f32x4 v = *reinterpret_cast<const f32x4*>(x + pix * C + 4 * c4);
        if (mask) v *= mask[pix];
        const int64_t p2 = ((int64_t)b * (H >> 1) + (yh >> 1)) * (W >> 1) + (xw >> 1);
        *reinterpret_cast<f32x4*>(x2 + p2 * (4 * C) + ((yh & 1) * 2 + (xw & 1)) * C + 4 * c4) = v;
    }
}
// dx[b][y][x][c] (+)= dx2[b][y/2][x/2][(y&1, x&1, c)] * mask
__global__ __launch_bounds__(256) void d2s_kernel(const float* __restrict__ dx2, const float* __restrict__ mask, float* __restrict__ dx,
                                                  int B, int H, int W, int C, int accumulate) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)B * H * W * C4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        int64_t pix = idx / C4;
        const int xw = (int)(pix % W);
        int64_t t = pix / W;
        const int yh = (int)(t % H);
        const int b = (int)(t / H);
        const int64_t p2 = ((int64_t)b * (H >> 1) + (yh >> 1)) * (W >> 1) + (xw >> 1);
        f32x4 v = *reinterpret_cast<const f32x4*>(dx2 + p2 * (4 * C) + ((yh & 1) * 2 + (xw & 1)) * C + 4 * c4);
        if (mask) v *= mask[pix];
        float* o = dx + pix * C + 4 * c4;
        if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = v;
    }
}
// to3 = 1: w2[co][ty][tx][(dy,dx,c)] = w[co][2ty+dy][2tx+dx][c] (0 where that tap does not exist);  to3 = 0: the inverse gather
__global__ __launch_bounds__(256) void w5x5_s2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int C, int to3) {
    const int64_t total = to3 ? (int64_t)Cout * 36 * C : (int64_t)Cout * 25 * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        int64_t t = idx / C;
        if (to3) {
            const int ph = (int)(t % 4), tap = (int)((t / 4) % 9), co = (int)(t / 36);
            const int ky = 2 * (tap / 3) + (ph >> 1), kx = 2 * (tap % 3) + (ph & 1);
            dst[idx] = (ky < 5 && kx < 5) ? src[((int64_t)co * 25 + ky * 5 + kx) * C + c] : 0.f;
        } else {
            const int k = (int)(t % 25), co = (int)(t / 25);
            const int ky = k / 5, kx = k % 5;
            dst[idx] = src[(((int64_t)co * 9 + (ky >> 1) * 3 + (kx >> 1)) * 4 + (ky & 1) * 2 + (kx & 1)) * C + c];
        }
    }
}
static bool s2d_ok(const TgConv* g) {
    static const bool off = getenv("TG_NO_WINO") != nullptr || getenv("TG_NO_S2D") != nullptr;
    if (off || (g->precision == TG_PREC_BF16 && getenv("TG_NO_WINO16"))) return false;
    return g->k == 5 && g->stride == 2 && g->pad == 2 && (g->H % 2) == 0 && (g->W % 2) == 0 && (g->Cin % 16) == 0 &&
           (g->Cout % 64) == 0 && g->Ho >= 32 && g->Wo >= 32 && g->Ho == g->H / 2 && g->Wo == g->W / 2;   // smaller: not worth 3 extra passes
}
static TgConv s2d_geom(const TgConv* g) {
    TgConv g2 = *g;
    g2.H = g->H / 2; g2.W = g->W / 2; g2.Cin = 4 * g->Cin; g2.k = 3; g2.stride = 1; g2.pad = 1;
    return g2;
}
static size_t s2d_x_floats(const TgConv* g) { return align_up((size_t)g->B * g->H * g->W * g->Cin, 64); }
static size_t s2d_w_floats(const TgConv* g) { return align_up((size_t)g->Cout * 36 * g->Cin, 64); }

// ---- forward ----------------------------------------------------------------------------------
extern "C" size_t tg_conv_fwd_ws_bytes(const TgConv* g) {
    if (!g) return 0;
    // room for up to 8 slabs of the output (plan_splits shrinks to what fits)
    size_t out = (size_t)g->B * g->Ho * g->Wo * g->Cout;
    size_t cap = (size_t)64 << 20;  // floats
    size_t want = out * 16;
    size_t base = ((want < cap ? want : cap) + conv_wino_floats(g)) * sizeof(float);
    if (s2d_ok(g)) {
        const TgConv g2 = s2d_geom(g);
        base = (s2d_x_floats(g) + s2d_w_floats(g)) * sizeof(float) + tg_conv_fwd_ws_bytes(&g2);
    }
    return base;
}

// Prepared weights.  Every weight rearrangement the conv kernels need (the Winograd transform U = G g Gt of the stride-1 3x3
// layers, the [Cin][taps][Cout] transpose of the gather dgrads, the 3x3 x 4C regrouping of the 5x5 stride-2 layers)
// depends on the weights only, so the caller may have it computed ONCE per optimiser step (once ever for the frozen VGG
// trunk) by tg_conv_wprep and hand it to tg_conv_fwd_p / tg_conv_dgrad_p.  `prep`: 0 = none given (prepare per call in
// the workspace, the tg_conv_fwd / tg_conv_dgrad behaviour), 1 = `wprep` is ready, -1 = fill `wprep` and return.
// (bf16 mode: wino16_kernel walks K in 16-channel steps)
static int wino_kc(const TgConv* g) {
    static const bool off16 = getenv("TG_NO_WINO16") != nullptr;
    return g->precision == TG_PREC_BF16 ? (off16 ? 0 : 16) : 8;
}
static bool wino_fwd_geom_ok(const TgConv* g) {
    static const bool off = getenv("TG_NO_WINO") != nullptr;
    const int kc = wino_kc(g);
    return !off && kc && g->k == 3 && g->stride == 1 && (g->Cin % kc) == 0 && (g->Cout % WINO_BN) == 0 && g->Ho >= 16 && g->Wo >= 16;
}
static bool wino_dgrad_geom_ok(const TgConv* g) {
    static const bool off = getenv("TG_NO_WINO") != nullptr;
    const int kc = wino_kc(g);
    return !off && kc && g->k == 3 && g->stride == 1 && (g->Cout % kc) == 0 && (g->Cin % WINO_BN) == 0 && g->H >= 16 && g->W >= 16;
}
static size_t dgrad_wt_floats(const TgConv* g);
extern "C" size_t tg_conv_wprep_bytes(const TgConv* g, int mode) {
    if (!g) return 0;
    if (s2d_ok(g)) {
        const TgConv g2 = s2d_geom(g);
        return s2d_w_floats(g) * sizeof(float) + tg_conv_wprep_bytes(&g2, mode);
    }
    if (mode == TG_WPREP_FWD) return wino_fwd_geom_ok(g) || wino22_fwd_geom_ok(g) ? conv_wino_floats(g) * sizeof(float) : 0;
    if (mode == TG_WPREP_DGRAD) return dgrad_wt_floats(g) * sizeof(float);
    return 0;
}

// what a forward call may ask of its launcher beyond the plain convolution (tg_conv_fwd_pool / _pool_code / _bnin)
struct FwdExtras {
    float* pool_dst = nullptr;              // 2x2 max-pool of the output, written from the output transform where the kernel can
    unsigned char* pool_code = nullptr;     // ... with the pool code, and dst itself NOT written
    const BnIn* in_bn = nullptr;            // the source is act(BN(src)), applied on load
    int pool_fused = 0;                     // out: the launcher wrote the pooled tensor
};
static int conv_fwd_impl(const TgConv* g, const float* x, const float* in_mask, const float* w, float* wprep, int prep,
                         const float* bias, const float* ratio, int act, float slope, float* y, float* ws, size_t ws_bytes,
                         tg_stream_t stream, float* pool_y = nullptr, FwdExtras* ex = nullptr) {
    if (pool_y) {
        // tg_conv_fwd_pool: the kernel that can write the pooled tensor from its output transform does (wino_pipe_kernel<.., POOL>);
        // every other route runs the pool kernel on y -- the same values either way (a maximum has no rounding)
        TG_REQUIRE(g && x && y && prep >= 0 && (g->Ho % 2) == 0 && (g->Wo % 2) == 0 && aligned16(pool_y),
                   "tg_conv_fwd_pool: even output sizes and a 16-byte aligned pool_y expected");
        int fused = 0;
        if (!s2d_ok(g) && !wino22_fwd_geom_ok(g)) {
            FwdExtras e;
            e.pool_dst = pool_y;
            const int rc1 = conv_fwd_impl(g, x, in_mask, w, wprep, prep, bias, ratio, act, slope, y, ws, ws_bytes, stream, nullptr, &e);
            if (rc1) return rc1;
            fused = e.pool_fused;
        } else if (int rc1 = conv_fwd_impl(g, x, in_mask, w, wprep, prep, bias, ratio, act, slope, y, ws, ws_bytes, stream)) {
            return rc1;
        }
        return fused ? TG_OK : tg_maxpool2_fwd(y, g->B, g->Ho, g->Wo, g->Cout, pool_y, stream);
    }
    int rc = check_conv(g, "tg_conv_fwd");
    if (rc) return rc;
    TG_REQUIRE(w && (prep < 0 || (x && y)), "tg_conv_fwd: null pointer");
    TG_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(wprep), "tg_conv_fwd: pointers must be 16-byte aligned");
    TG_REQUIRE(prep == 0 || wprep != nullptr, "tg_conv_fwd: prepared weights expected");
    const bool ws_ok = ws && aligned16(ws) && ws_bytes >= tg_conv_fwd_ws_bytes(g);
    if (s2d_ok(g) && (prep < 0 || ws_ok)) {
        // 5x5 stride 2 -> 3x3 stride 1 over the space-to-depth input (see s2d_kernel)
        hipStream_t s = (hipStream_t)stream;
        const TgConv g2 = s2d_geom(g);
        float* x2 = ws;
        float* w2 = prep ? wprep : ws + s2d_x_floats(g);
        float* ws2 = ws ? ws + s2d_x_floats(g) + s2d_w_floats(g) : nullptr;
        if (prep < 0 && g_wprep_capture) {
            // batched preparation: the inner 3x3 problem describes its transform (on the regrouped weights), which is then
            // re-pointed at the 5x5 weights themselves (kind 4 gathers them on the fly; the regrouped copy is not needed)
            const int rc2 = conv_fwd_impl(&g2, nullptr, nullptr, w2, wprep + s2d_w_floats(g), -1, nullptr, nullptr, 0, 0.f, nullptr, nullptr,
                                          0, stream);
            if (rc2 == TG_OK && g_wprep_captured == 1 && g_wprep_capture->kind == 1) {
                g_wprep_capture->kind = 4; g_wprep_capture->w = w; g_wprep_capture->taps = g->Cin;
            } else {
                g_wprep_captured = -1;
            }
            return rc2;
        }
        if (prep <= 0) {
            hipLaunchKernelGGL(w5x5_s2d_kernel, dim3(ew_grid((int64_t)g->Cout * 36 * g->Cin, 256)), dim3(256), 0, s, w, w2, g->Cout, g->Cin, 1);
            TG_CHECK_LAUNCH("w5x5_s2d_kernel");
        }
        if (prep >= 0) {
            hipLaunchKernelGGL(s2d_kernel, dim3(ew_grid((int64_t)g->B * g->H * g->W * (g->Cin / 4), 256)), dim3(256), 0, s, x, in_mask, x2,
                               g->B, g->H, g->W, g->Cin);
            TG_CHECK_LAUNCH("s2d_kernel");
        }
        AlgScale sc(25.0 / 36.0);
        return conv_fwd_impl(&g2, x2, nullptr, w2, prep ? wprep + s2d_w_floats(g) : nullptr, prep, bias, ratio, act, slope, y, ws2,
                             ws ? ws_bytes - (s2d_x_floats(g) + s2d_w_floats(g)) * sizeof(float) : 0, stream);
    }
    TG_REQUIRE(!(s2d_ok(g) && prep > 0), "tg_conv_fwd: workspace too small for the prepared 5x5 stride-2 path");
    IGemmParams p = {};
    p.src = x; p.amask = in_mask; p.wmat = w; p.bias = bias; p.rowscale = ratio; p.dst = y; p.ws = ws;
    p.B = g->B; p.IH = g->H; p.IW = g->W; p.C = g->Cin;
    p.OH = g->Ho; p.OW = g->Wo; p.N = g->Cout; p.M = g->B * g->Ho * g->Wo;
    p.DH = g->Ho; p.DW = g->Wo; p.ds = 1; p.dy0 = 0; p.dx0 = 0;
    p.TH = g->k; p.TW = g->k; p.ss = g->stride; p.tstep = 1; p.sy0 = -g->pad; p.sx0 = -g->pad;
    p.KW = g->k; p.kstep = 1; p.ky0 = 0; p.kx0 = 0;
    p.Kfull = g->k * g->k * g->Cin;
    p.act = act; p.slope = slope; p.accumulate = 0;
    p.bf16 = g->precision == TG_PREC_BF16;
    p.wino4 = g->precision == TG_PREC_F32_WINO4;
    p.Ktot = p.TH * p.TW * p.C; p.nchunks = cdiv(p.C, 32);
    p.T = (p.C % 4) ? cdiv(p.Ktot, 32) : p.TH * p.TW * p.nchunks;
    size_t ws_floats = ws ? ws_bytes / sizeof(float) : 0;
    const size_t uf = conv_wino_floats(g);
    if (wino22_fwd_geom_ok(g) && (prep || (ws_floats >= uf && aligned16(ws)))) {
        // 4x4 stride 2: Winograd F(2x2,2x2) over the shifted space-to-depth view, gathered on the fly (wino22.inc)
        p.w_raw = w;
        p.wino_u = prep ? wprep : ws;
        p.wino_ready = prep;
        if (!prep) {
            p.ws = ws + uf;
            ws_floats -= uf;
        }
        return launch_wino22(g, p, 0, ws_floats, (hipStream_t)stream);
    }
    if (wino_fwd_geom_ok(g) && (prep || (ws_floats >= uf && aligned16(ws)))) {
        // transformed weights: prepared by the caller, or at the head of the workspace
        p.w_raw = w; p.w_sn = (long)p.Kfull; p.w_sk = 1; p.w_stap = g->Cin;
        p.wino_u = prep ? wprep : ws;
        p.wino_ready = prep;
        bool in_ws = !prep;
        if (wino44_prepared_unusable(p)) {
            // `wprep` holds the F(4x4,3x3) image (its layout follows the geometry alone) and this launch -- row scales, an input
            // mask, or a batch whose output reaches 2 GB -- runs on wino_kernel: F(2x2,3x3) weights, transformed per call
            TG_REQUIRE(ws_floats >= uf && aligned16(ws), "tg_conv_fwd: workspace too small to re-prepare the weights (%zu < %zu floats)",
                       ws_floats, uf);
            p.wino_u = ws;
            p.wino_ready = 0;
            in_ws = true;
        }
        TG_REQUIRE(p.bf16 ? wino16_ok(p) : wino_ok(p), "tg_conv_fwd: internal: Winograd geometry predicate mismatch");
        if (in_ws) {
            p.ws = ws + uf;
            ws_floats -= uf;
        }
    }
    if (prep < 0 && !p.wino_u) return TG_OK;              // this layer runs on the raw weights: nothing to prepare
    if (ex) {
        p.pool_dst = ex->pool_dst;
        p.pool_code = ex->pool_code;
        p.pool_only = ex->pool_code != nullptr;
        if (ex->in_bn) {
            p.in_bn = *ex->in_bn;
            TG_REQUIRE(smallconv_bnin_fwd_ok(p), "tg_conv_fwd_bnin: geometry not supported (ask tg_conv_bnin_supported first)");
        }
    }
    plan_splits(p, ws_floats);
    rc = launch_igemm(p, (hipStream_t)stream, ws_floats);
    if (ex) ex->pool_fused = p.pool_done;
    return rc;
}
extern "C" int tg_conv_fwd_pool(const TgConv* g, const float* x, const float* in_mask, const float* w, const float* wprep,
                                const float* bias, const float* ratio, int act, float slope, float* y, float* pool_y, float* ws,
                                size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(pool_y, "tg_conv_fwd_pool: null pointer");
    return conv_fwd_impl(g, x, in_mask, w, const_cast<float*>(wprep), wprep ? 1 : 0, bias, ratio, act, slope, y, ws, ws_bytes, stream,
                         pool_y);
}
// conv -> ReLU -> 2x2 max-pool where the full-resolution conv output has no reader but the pool and the pool's backward
// (VGG16 features[2..4], [7..9]): pooled tensor + one code byte per pooled element (IGemmParams::pool_code), the conv output
// itself is never written.  Only where wino_pipe_kernel<.., POOL> takes the launch in one K split.
extern "C" int tg_conv_pool_code_supported(const TgConv* g) {
    if (!g || (g->precision != TG_PREC_F32 && g->precision != TG_PREC_BF16) || !wino_fwd_geom_ok(g) || s2d_ok(g) || (g->Ho & 1) || (g->Wo & 1)) return 0;
    if (getenv("TG_NO_FUSED_POOL") || getenv("TG_NO_POOL_CODE") || getenv("TG_WINO_NO_PIPE") || getenv("TG_WINO_NO_FAST")) return 0;
    if ((size_t)g->B * g->Ho * g->Wo * g->Cout * 4 >= ((size_t)1 << 31) || (size_t)g->B * g->H * g->W * g->Cin * 4 >= ((size_t)1 << 31)) return 0;
    const int nchunks = g->Cin / wino_kc(g);            // K steps of 8 (fp32) / 16 (bf16 operands) channels
    if (nchunks < 2) return 0;
    const long work = (long)cdiv(g->Wo, 16) * cdiv(g->Ho, 16) * g->B * (g->Cout / WINO_BN);
    return nchunks < 16 || work >= 4L * WINO_PLAN_CUS ? 1 : 0;         // (launch_wino: one K split)
}
extern "C" int tg_conv_fwd_pool_code(const TgConv* g, const float* x, const float* w, const float* wprep, const float* bias,
                                     float* pool_y, unsigned char* code, float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(pool_y && code && aligned16(pool_y) && (reinterpret_cast<uintptr_t>(code) & 3) == 0, "tg_conv_fwd_pool_code: bad pointers");
    TG_REQUIRE(tg_conv_pool_code_supported(g), "tg_conv_fwd_pool_code: geometry not supported (ask tg_conv_pool_code_supported first)");
    FwdExtras e;
    e.pool_dst = pool_y;
    e.pool_code = code;
    // (dst is not written: the pooled buffer stands in for the pointer checks)
    const int rc = conv_fwd_impl(g, x, nullptr, w, const_cast<float*>(wprep), wprep ? 1 : 0, bias, nullptr, TG_ACT_RELU, 0.f, pool_y, ws,
                                 ws_bytes, stream, nullptr, &e);
    if (rc) return rc;
    TG_REQUIRE(e.pool_fused, "tg_conv_fwd_pool_code: internal: the launch did not take the fused path");
    return TG_OK;
}
static bool bnin_geom_ok(const TgConv* g) {
    return g && g->Cout == 1 && g->Cin == 64 && g->k == 3 && g->stride == 1 && g->pad == 1 && (g->Wo % 4) == 0;
}
static BnIn bnin_of(const TgBnAct* bn) { return BnIn{bn->mean, bn->rstd, bn->gamma, bn->beta, bn->act, bn->slope}; }
extern "C" int tg_conv_bnin_supported(const TgConv* g, int wgrad) {
    if (!bnin_geom_ok(g)) return 0;
    if (wgrad) {
        WgradParams p = {};
        p.B = g->B; p.H = g->H; p.W = g->W; p.C = g->Cin; p.Ho = g->Ho; p.Wo = g->Wo; p.Cout = g->Cout;
        p.k = g->k; p.stride = g->stride; p.pad = g->pad;
        p.Mpix = g->B * g->Ho * g->Wo; p.Ktot = g->k * g->k * g->Cin;
        return smallconv_wgrad_applies(p) && smallconv_bnin_wgrad_ok(p) ? 1 : 0;
    }
    IGemmParams p = {};
    p.B = g->B; p.IH = g->H; p.IW = g->W; p.C = g->Cin; p.OH = g->Ho; p.OW = g->Wo; p.N = g->Cout; p.M = g->B * g->Ho * g->Wo;
    p.TH = g->k; p.TW = g->k; p.ss = g->stride; p.tstep = 1; p.sy0 = -g->pad; p.sx0 = -g->pad;
    return smallconv_bnin_fwd_ok(p) ? 1 : 0;
}
extern "C" int tg_conv_fwd_bnin(const TgConv* g, const float* x, const TgBnAct* bn, const float* w, const float* bias, int act,
                                float slope, float* y, float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(bn && bn->mean && bn->rstd && bn->gamma && bn->beta, "tg_conv_fwd_bnin: null pointer");
    TG_REQUIRE(tg_conv_bnin_supported(g, 0), "tg_conv_fwd_bnin: geometry not supported (ask tg_conv_bnin_supported first)");
    const BnIn b = bnin_of(bn);
    FwdExtras e;
    e.in_bn = &b;
    return conv_fwd_impl(g, x, nullptr, w, nullptr, 0, bias, nullptr, act, slope, y, ws, ws_bytes, stream, nullptr, &e);
}
extern "C" int tg_conv_fwd(const TgConv* g, const float* x, const float* in_mask, const float* w, const float* bias,
                           const float* ratio, int act, float slope, float* y, float* ws, size_t ws_bytes,
                           tg_stream_t stream) {
    return conv_fwd_impl(g, x, in_mask, w, nullptr, 0, bias, ratio, act, slope, y, ws, ws_bytes, stream);
}
extern "C" int tg_conv_fwd_p(const TgConv* g, const float* x, const float* in_mask, const float* w, const float* wprep,
                             const float* bias, const float* ratio, int act, float slope, float* y, float* ws, size_t ws_bytes,
                             tg_stream_t stream) {
    return conv_fwd_impl(g, x, in_mask, w, const_cast<float*>(wprep), wprep ? 1 : 0, bias, ratio, act, slope, y, ws, ws_bytes,
                         stream);
}

// ---- dgrad ---------------------------------------------------------------------------------------
// [Cout][T][Cin] -> [Cin][T][Cout]
__global__ __launch_bounds__(256) void transpose_w_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                          int cout, int taps, int cin) {
    const size_t total = (size_t)cout * taps * cin;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int co = (int)(idx % cout);
        size_t r = idx / cout;
        int t = (int)(r % taps);
        int ci = (int)(r / taps);
        wt[idx] = w[((size_t)co * taps + t) * cin + ci];
    }
}

// head of the dgrad workspace: the transposed weights, or (stride-1 3x3) the Winograd-transformed ones
static size_t dgrad_wt_floats(const TgConv* g) {
    size_t wt = align_up((size_t)g->Cout * g->k * g->k * g->Cin, 64);
    size_t uf = conv_wino_floats(g);
    return wt > uf ? wt : uf;
}
extern "C" size_t tg_conv_dgrad_ws_bytes(const TgConv* g) {
    if (!g) return 0;
    size_t wt = dgrad_wt_floats(g);
    size_t out = (size_t)g->B * g->H * g->W * g->Cin;
    size_t cap = (size_t)64 << 20;
    size_t want = out * 16;
    if (s2d_ok(g)) {
        const TgConv g2 = s2d_geom(g);
        return (s2d_x_floats(g) + s2d_w_floats(g)) * sizeof(float) + tg_conv_dgrad_ws_bytes(&g2);
    }
    return (wt + (want < cap ? want : cap)) * sizeof(float);
}

static int conv_dgrad_impl(const TgConv* g, const float* dy, const float* w, float* wprep, int prep, const float* in_mask, float* dx,
                           int accumulate, const float* gate, int gate_act, float gate_slope, float* ws, size_t ws_bytes,
                           tg_stream_t stream);
extern "C" int tg_conv_dgrad(const TgConv* g, const float* dy, const float* w, const float* in_mask, float* dx,
                             int accumulate, float* ws, size_t ws_bytes, tg_stream_t stream) {
    return conv_dgrad_impl(g, dy, w, nullptr, 0, in_mask, dx, accumulate, nullptr, 0, 0.f, ws, ws_bytes, stream);
}
extern "C" int tg_conv_dgrad_gated(const TgConv* g, const float* dy, const float* w, const float* in_mask, const float* x_act,
                                   int act, float slope, float* dx, float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(x_act != nullptr && (act == TG_ACT_RELU || act == TG_ACT_LEAKY), "tg_conv_dgrad_gated: needs x_act and a ReLU/LeakyReLU");
    return conv_dgrad_impl(g, dy, w, nullptr, 0, in_mask, dx, 0, x_act, act, slope, ws, ws_bytes, stream);
}
// x_act == NULL: plain dgrad (accumulate honoured); else the gated form (accumulate must be 0)
extern "C" int tg_conv_dgrad_p(const TgConv* g, const float* dy, const float* w, const float* wprep, const float* in_mask,
                               const float* x_act, int act, float slope, float* dx, int accumulate, float* ws, size_t ws_bytes,
                               tg_stream_t stream) {
    TG_REQUIRE(x_act == nullptr || ((act == TG_ACT_RELU || act == TG_ACT_LEAKY) && !accumulate),
               "tg_conv_dgrad_p: a gated dgrad needs a ReLU/LeakyReLU and accumulate == 0");
    // a gated dgrad of a 5x5 stride-2 layer does not take the space-to-depth path its prepared weights were laid out for
    const bool usable = wprep != nullptr && !(x_act != nullptr && g && s2d_ok(g));
    return conv_dgrad_impl(g, dy, w, usable ? const_cast<float*>(wprep) : nullptr, usable ? 1 : 0, in_mask, dx, accumulate, x_act,
                           act, slope, ws, ws_bytes, stream);
}
extern "C" int tg_conv_wprep(const TgConv* g, int mode, const float* w, float* wprep, tg_stream_t stream) {
    TG_REQUIRE(g && w && wprep, "tg_conv_wprep: null pointer");
    TG_REQUIRE(mode == TG_WPREP_FWD || mode == TG_WPREP_DGRAD, "tg_conv_wprep: bad mode %d", mode);
    TG_REQUIRE(tg_conv_wprep_bytes(g, mode) > 0, "tg_conv_wprep: this (geometry, mode) runs on the raw weights");
    if (mode == TG_WPREP_FWD)
        return conv_fwd_impl(g, nullptr, nullptr, w, wprep, -1, nullptr, nullptr, 0, 0.f, nullptr, nullptr, 0, stream);
    return conv_dgrad_impl(g, nullptr, w, wprep, -1, nullptr, nullptr, 0, nullptr, 0, 0.f, nullptr, 0, stream);
}
// Batched weight preparation.  tg_conv_wprep_item describes what tg_conv_wprep(g, mode, w, wprep) would launch as one POD
// descriptor (returns 1), or returns 0 when this (geometry, mode) is not batchable (two dependent passes, bf16 / F(2x2,2x2)
// transforms) or needs no preparation; the caller keeps the descriptors of all its layers in ONE device array and has them
// executed by ONE launch per optimiser step (tg_conv_wprep_run) instead of one launch per layer and mode.
extern "C" size_t tg_conv_wprep_item_bytes(void) { return sizeof(TgWprepItem); }
extern "C" int tg_conv_wprep_item(const TgConv* g, int mode, const float* w, float* wprep, void* item_out) {
    if (!g || !w || !wprep || !item_out || (mode != TG_WPREP_FWD && mode != TG_WPREP_DGRAD)) return 0;
    if (tg_conv_wprep_bytes(g, mode) == 0) return 0;
    TgWprepItem it = {};
    g_wprep_capture = &it;
    g_wprep_captured = 0;
    int rc = mode == TG_WPREP_FWD ? conv_fwd_impl(g, nullptr, nullptr, w, wprep, -1, nullptr, nullptr, 0, 0.f, nullptr, nullptr, 0, nullptr)
                                  : conv_dgrad_impl(g, nullptr, w, wprep, -1, nullptr, nullptr, 0, nullptr, 0, 0.f, nullptr, 0, nullptr);
    g_wprep_capture = nullptr;
    if (rc != TG_OK || g_wprep_captured != 1) return 0;
    memcpy(item_out, &it, sizeof(it));
    return 1;
}
extern "C" int tg_conv_wprep_run(const void* items_dev, int n, tg_stream_t stream) {
    TG_REQUIRE(items_dev && n > 0, "tg_conv_wprep_run: bad arguments");
    // 2048 x 256 threads per descriptor: the large layers need that many to cover the memory latency, the small ones' surplus
    // blocks find nothing to do and leave
    hipLaunchKernelGGL(wprep_multi_kernel, dim3(2048, n), dim3(256), 0, (hipStream_t)stream, static_cast<const TgWprepItem*>(items_dev));
    TG_CHECK_LAUNCH("wprep_multi_kernel");
    return TG_OK;
}
static int conv_dgrad_impl(const TgConv* g, const float* dy, const float* w, float* wprep, int prep, const float* in_mask, float* dx,
                           int accumulate, const float* gate, int gate_act, float gate_slope, float* ws, size_t ws_bytes,
                           tg_stream_t stream) {
    int rc = check_conv(g, "tg_conv_dgrad");
    if (rc) return rc;
    TG_REQUIRE(w && (prep < 0 || (dy && dx && ws)), "tg_conv_dgrad: null pointer");
    TG_REQUIRE(aligned16(dy) && aligned16(w) && aligned16(dx) && aligned16(ws) && aligned16(wprep),
               "tg_conv_dgrad: pointers must be 16-byte aligned");
    TG_REQUIRE(prep == 0 || wprep != nullptr, "tg_conv_dgrad: prepared weights expected");
    if (s2d_ok(g) && gate == nullptr && (prep < 0 || ws_bytes >= tg_conv_dgrad_ws_bytes(g))) {
        // dx2 = 3x3 stride-1 dgrad over the space-to-depth layout, then depth-to-space (+ mask, + accumulate)
        hipStream_t s2 = (hipStream_t)stream;
        const TgConv g2 = s2d_geom(g);
        float* dx2 = ws;
        float* w2 = prep ? wprep : ws + s2d_x_floats(g);
        float* wsr = ws ? ws + s2d_x_floats(g) + s2d_w_floats(g) : nullptr;
        if (prep < 0 && g_wprep_capture) {
            const int rc2 = conv_dgrad_impl(&g2, nullptr, w2, wprep + s2d_w_floats(g), -1, nullptr, nullptr, 0, nullptr, 0, 0.f, nullptr, 0, stream);
            if (rc2 == TG_OK && g_wprep_captured == 1 && g_wprep_capture->kind == 1) {
                g_wprep_capture->kind = 4; g_wprep_capture->w = w; g_wprep_capture->taps = g->Cin;
            } else {
                g_wprep_captured = -1;
            }
            return rc2;
        }
        if (prep <= 0) {
            hipLaunchKernelGGL(w5x5_s2d_kernel, dim3(ew_grid((int64_t)g->Cout * 36 * g->Cin, 256)), dim3(256), 0, s2, w, w2, g->Cout, g->Cin, 1);
            TG_CHECK_LAUNCH("w5x5_s2d_kernel");
        }
        {
            AlgScale sc(25.0 / 36.0);
            rc = conv_dgrad_impl(&g2, dy, w2, prep ? wprep + s2d_w_floats(g) : nullptr, prep, nullptr, dx2, 0, nullptr, 0, 0.f, wsr,
                                 ws ? ws_bytes - (s2d_x_floats(g) + s2d_w_floats(g)) * sizeof(float) : 0, stream);
        }
        if (rc || prep < 0) return rc;
        hipLaunchKernelGGL(d2s_kernel, dim3(ew_grid((int64_t)g->B * g->H * g->W * (g->Cin / 4), 256)), dim3(256), 0, s2, dx2, in_mask, dx,
                           g->B, g->H, g->W, g->Cin, accumulate);
        TG_CHECK_LAUNCH("d2s_kernel");
        return TG_OK;
    }
    TG_REQUIRE(!(s2d_ok(g) && gate == nullptr && prep > 0), "tg_conv_dgrad: workspace too small for the prepared 5x5 stride-2 path");
    const int taps = g->k * g->k;
    // head of the workspace = transposed / transformed weights unless the caller prepared them
    const size_t wt_floats = prep ? 0 : dgrad_wt_floats(g);
    TG_REQUIRE(prep < 0 || ws_bytes >= wt_floats * sizeof(float), "tg_conv_dgrad: workspace too small (%zu < %zu)", ws_bytes,
               wt_floats * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    float* wt = prep ? wprep : ws;
    float* ws2 = ws ? ws + wt_floats : nullptr;
    const size_t ws2_floats = ws ? ws_bytes / sizeof(float) - wt_floats : 0;
    if (wino_dgrad_geom_ok(g)) {
        // stride-1 3x3: the Winograd kernel reads W[co][tap][ci] through strides (n = ci, k = co); no transposed copy
        IGemmParams p = {};
        p.src = dy; p.rowscale = in_mask; p.dst = dx; p.ws = ws2;
        p.B = g->B; p.IH = g->Ho; p.IW = g->Wo; p.C = g->Cout;
        p.OH = g->H; p.OW = g->W; p.N = g->Cin; p.M = g->B * g->H * g->W;
        p.DH = g->H; p.DW = g->W; p.ds = 1;
        p.ky0 = 0; p.kx0 = 0; p.TH = 3; p.TW = 3; p.ss = 1; p.tstep = -1; p.sy0 = g->pad; p.sx0 = g->pad;
        p.KW = 3; p.kstep = 1; p.Kfull = taps * g->Cout;
        p.act = TG_ACT_NONE; p.accumulate = accumulate;
        p.gate = gate; p.gate_act = gate_act; p.gate_slope = gate_slope;
        p.bf16 = g->precision == TG_PREC_BF16;
        p.w_raw = w; p.w_sn = 1; p.w_sk = (long)taps * g->Cin; p.w_stap = g->Cin;
        p.wino_u = wt;
        p.wino_ready = prep;
        p.wino4 = g->precision == TG_PREC_F32_WINO4;
        if (wino44_ok(p)) return launch_wino44(p, s);
        size_t avail = ws2_floats;
        if (wino44_prepared_unusable(p)) {         // see conv_fwd_impl: F(4x4) image prepared, F(2x2) launch -> transform per call
            const size_t uf = dgrad_wt_floats(g);
            TG_REQUIRE(ws && aligned16(ws) && avail >= uf, "tg_conv_dgrad: workspace too small to re-prepare the weights (%zu < %zu floats)",
                       avail, uf);
            p.wino_u = ws;
            p.wino_ready = 0;
            p.ws = ws + uf;
            avail -= uf;
        }
        TG_REQUIRE(p.bf16 ? wino16_ok(p) : wino_ok(p), "tg_conv_dgrad: internal: Winograd geometry predicate mismatch");
        return p.bf16 ? launch_wino16(p, avail, s) : launch_wino(p, avail, s);
    }
    if (wino22_dgrad_geom_ok(g)) {
        // 4x4 stride 2: the four parity classes as Winograd F(2x2,2x2) problems in one launch (wino22.inc)
        IGemmParams p = {};
        p.src = dy; p.rowscale = in_mask; p.dst = dx; p.ws = ws2;
        p.act = TG_ACT_NONE; p.accumulate = accumulate;
        p.gate = gate; p.gate_act = gate_act; p.gate_slope = gate_slope;
        p.w_raw = w;
        p.wino_u = wt;
        p.wino_ready = prep;
        return launch_wino22(g, p, 1, ws2_floats, s);
    }
    if (prep < 0 && g_wprep_capture) {
        TgWprepItem it = {};
        it.kind = 2; it.N = g->Cout; it.K = g->Cin; it.taps = taps; it.w = w; it.out = wt;
        *g_wprep_capture = it;
        g_wprep_captured = 1;
        return TG_OK;
    }
    if (prep <= 0) {
        hipLaunchKernelGGL(transpose_w_kernel, dim3(ew_grid((int64_t)g->Cout * taps * g->Cin, 256)), dim3(256), 0, s, w, wt,
                           g->Cout, taps, g->Cin);
        TG_CHECK_LAUNCH("transpose_w_kernel");
    }
    if (prep < 0) return TG_OK;

    const int st = g->stride;
    IGemmParams cls[4];
    int ncls = 0;
    const bool mergeable = st * st <= 4;
    for (int py = 0; py < st; ++py) {
        for (int px = 0; px < st; ++px) {
            IGemmParams p = {};
            p.src = dy; p.amask = nullptr; p.wmat = wt; p.bias = nullptr; p.rowscale = in_mask; p.dst = dx; p.ws = ws2;
            p.B = g->B; p.IH = g->Ho; p.IW = g->Wo; p.C = g->Cout;
            p.OH = (g->H - py + st - 1) / st; p.OW = (g->W - px + st - 1) / st;
            if (p.OH <= 0 || p.OW <= 0) continue;
            p.N = g->Cin; p.M = g->B * p.OH * p.OW;
            p.DH = g->H; p.DW = g->W; p.ds = st; p.dy0 = py; p.dx0 = px;
            p.ky0 = (py + g->pad) % st; p.kx0 = (px + g->pad) % st;
            p.TH = p.ky0 < g->k ? (g->k - p.ky0 + st - 1) / st : 0;
            p.TW = p.kx0 < g->k ? (g->k - p.kx0 + st - 1) / st : 0;
            p.ss = 1; p.tstep = -1;
            p.sy0 = (py + g->pad - p.ky0) / st; p.sx0 = (px + g->pad - p.kx0) / st;
            p.KW = g->k; p.kstep = st;
            p.Kfull = taps * g->Cout;
            p.act = TG_ACT_NONE; p.slope = 0.f; p.accumulate = accumulate;
            p.gate = gate; p.gate_act = gate_act; p.gate_slope = gate_slope;
            p.bf16 = g->precision == TG_PREC_BF16;
            p.Ktot = p.TH * p.TW * p.C; p.nchunks = cdiv(p.C, 32);
            p.T = (p.C % 4) ? cdiv(p.Ktot, 32) : p.TH * p.TW * p.nchunks;
            if (mergeable) {
                cls[ncls++] = p;
            } else {
                plan_splits(p, ws2_floats);
                rc = launch_igemm(p, s, ws2_floats);
                if (rc) return rc;
            }
        }
    }
    if (mergeable && ncls > 0) {
        // the parity classes of a stride-2 dgrad go out as ONE patch-kernel launch when they qualify
        if (smallconv_to1_multi_applies(cls, ncls)) {
            // 64 -> 1 channel (D's first layer): the four classes in one launch read dy once from memory (they walk the same
            // pixels at the same time) instead of four times
            double fl = 0.0, by = 0.0;
            for (int i = 0; i < ncls; ++i) {
                cls[i].Ktot = cls[i].TH * cls[i].TW * cls[i].C;
                fl += 2.0 * cls[i].M * (double)cls[i].N * cls[i].Ktot;
                by += 4.0 * ((double)cls[i].M * cls[i].N + (double)cls[i].N * cls[i].Ktot);
            }
            by += 4.0 * (double)g->B * g->Ho * g->Wo * g->Cout;
            ProfScope ps(s, 2, fl, by, cls[0].M * ncls, 1, cls[0].TH * cls[0].TW * cls[0].C, cls[0].C, 1, 2004);
            return smallconv_to1_multi_launch(cls, ncls, s);
        }
        if (ncls > 1 && !smallconv_fwd_applies(cls[0])) {
            IGemmParams tmp[4];
            for (int i = 0; i < ncls; ++i) tmp[i] = cls[i];
            if (try_pgemm(tmp, ncls, ws2_floats, s, &rc)) return rc;
            for (int i = 0; i < ncls; ++i) tmp[i] = cls[i];
            if (try_igemm_multi(tmp, ncls, ws2, ws2_floats, s, &rc)) return rc;
        }
        for (int i = 0; i < ncls; ++i) {
            plan_splits(cls[i], ws2_floats);
            rc = launch_igemm(cls[i], s, ws2_floats);
            if (rc) return rc;
        }
    }
    return TG_OK;
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[cout][(tap,c)] = sum over pixels; both operands pixel-major ([k][m] LDS images, ds_read_b32)
// ------------------------------------------------------------------------------------------------

template <int WAVES_M, int WAVES_N, int WM, int WN, bool SCALAR_A, bool SCALAR_B>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, BK = 32;
    constexpr int A_V = BK * BM / 4 / 256;  // float4 loads per thread (vector path)
    constexpr int B_V = BK * BN / 4 / 256;
    constexpr int A_S = BK * BM / 256;      // scalar loads per thread
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(BN == 128, "column mapping assumes BN == 128");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                // [2][BK][BM]
    float* Bs = smem + 2 * BK * BM;  // [2][BK][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // flat grid, XCD-remapped: the N'/Cout tiles of one pixel range (split) run on one XCD and share its L2
    const int work = xcd_remap(blockIdx.x, p.nx * p.ny * p.splits);
    const int bz = work / (p.nx * p.ny);
    const int bxy = work - bz * (p.nx * p.ny);
    const int n0 = (bxy % p.nx) * BN, c0m = (bxy / p.nx) * BM;
    const int t_begin = bz * p.steps_per_split;
    const int t_end = min(p.T, t_begin + p.steps_per_split);

    // B' columns owned by this thread: n = n0 + 4*(tid&31) + e  ->  (tap, c), fixed over the K loop
    int bky[4], bkx[4], bc[4];
    bool bv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int n = n0 + 4 * (tid & 31) + e;
        bv[e] = n < p.Ktot;
        int nn = bv[e] ? n : 0;
        int tap = nn / p.C;
        bc[e] = nn - tap * p.C;
        bky[e] = tap / p.k - p.pad;
        bkx[e] = tap % p.k - p.pad;
    }

    f32x4 ra[SCALAR_A ? 1 : A_V];
    float ras[SCALAR_A ? A_S : 1];
    f32x4 rbv[B_V];
    // vector paths: the source-mask value travels with its quad and is multiplied in at the LDS store (see igemm_body)
    float rbm[B_V];

    // (b, oy, ox) of the B' rows this thread gathers, for the NEXT K step; advanced by 32 pixels per step
    int pb[B_V], py[B_V], px[B_V];
#pragma unroll
    for (int i = 0; i < B_V; ++i) {
        int m = t_begin * BK + (tid >> 5) + 8 * i;
        px[i] = m % p.Wo;
        int tt = m / p.Wo;
        py[i] = tt % p.Ho;
        pb[i] = tt / p.Ho;
    }
    // row-segment mode: the step's 32 pixels share (b, oy); wave-uniform counters, advanced incrementally
    int rs_ox0 = (t_begin * BK) % p.Wo, rs_oy = ((t_begin * BK) / p.Wo) % p.Ho, rs_b = (t_begin * BK) / p.Wo / p.Ho;

    auto gload = [&](int t) {
        const int mbase = t * BK;
        // A' = dy rows
        if constexpr (!SCALAR_A) {
            constexpr int PER_ROW = BM / 4;
#pragma unroll
            for (int i = 0; i < A_V; ++i) {
                int idx = tid + 256 * i;
                int krow = idx / PER_ROW, c4 = idx % PER_ROW;
                int m = mbase + krow, co = c0m + 4 * c4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < p.Mpix && co < p.Cout) v = *reinterpret_cast<const f32x4*>(p.dy + (size_t)m * p.Cout + co);
                ra[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_S; ++i) {
                int idx = tid + 256 * i;
                int krow = idx / BM, col = idx % BM;
                int m = mbase + krow, co = c0m + col;
                ras[i] = (m < p.Mpix && co < p.Cout) ? p.dy[(size_t)m * p.Cout + co] : 0.f;
            }
        }
        // B' = tap-shifted, masked input rows
        if (!SCALAR_B && p.rowseg) {
            const int iy = rs_oy * p.stride + bky[0];
            const bool vy = bv[0] && iy >= 0 && iy < p.H && mbase < p.Mpix;
            const size_t rowoff = ((size_t)rs_b * p.H + (vy ? iy : 0)) * p.W;
            const int ixb = rs_ox0 * p.stride + bkx[0];
#pragma unroll
            for (int i = 0; i < B_V; ++i) {
                const int ix = ixb + ((tid >> 5) + 8 * i) * p.stride;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                float mk = 1.f;
                if (vy && ix >= 0 && ix < p.W) {
                    const size_t pix = rowoff + ix;
                    v = *reinterpret_cast<const f32x4*>(p.x + pix * p.C + bc[0]);
                    if (p.amask) mk = p.amask[pix];
                }
                rbv[i] = v;
                rbm[i] = mk;
            }
            rs_ox0 += BK;
            if (rs_ox0 >= p.Wo) {
                rs_ox0 = 0;
                if (++rs_oy == p.Ho) { rs_oy = 0; ++rs_b; }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < B_V; ++i) {
            int krow = (tid >> 5) + 8 * i;
            int m = mbase + krow;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!SCALAR_B) {
                const int ox = px[i], oy = py[i], b = pb[i];
                const int iy = oy * p.stride + bky[0], ix = ox * p.stride + bkx[0];
                float mk = 1.f;
                if (m < p.Mpix && bv[0] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                    const size_t pix = ((size_t)b * p.H + iy) * p.W + ix;
                    v = *reinterpret_cast<const f32x4*>(p.x + pix * p.C + bc[0]);
                    if (p.amask) mk = p.amask[pix];
                }
                rbm[i] = mk;
            } else if (m < p.Mpix) {
                const int ox = px[i], oy = py[i], b = pb[i];
                {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int iy = oy * p.stride + bky[e], ix = ox * p.stride + bkx[e];
                        if (bv[e] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                            size_t pix = ((size_t)b * p.H + iy) * p.W + ix;
                            float sv = p.x[pix * p.C + bc[e]];
                            if (p.amask) sv *= p.amask[pix];
                            v[e] = sv;
                        }
                    }
                }
            }
            rbv[i] = v;
            // advance this row by BK pixels for the next step
            px[i] += BK;
            while (px[i] >= p.Wo) {
                px[i] -= p.Wo;
                if (++py[i] == p.Ho) { py[i] = 0; ++pb[i]; }
            }
        }
    };
    auto sstore = [&](int buf) {
        float* Ab = As + buf * BK * BM;
        float* Bb = Bs + buf * BK * BN;
        if constexpr (!SCALAR_B) {
            if (p.amask) {
#pragma unroll
                for (int i = 0; i < B_V; ++i) rbv[i] *= rbm[i];
            }
        }
        if constexpr (!SCALAR_A) {
            constexpr int PER_ROW = BM / 4;
#pragma unroll
            for (int i = 0; i < A_V; ++i) {
                int idx = tid + 256 * i;
                int krow = idx / PER_ROW, c4 = idx % PER_ROW;
                *reinterpret_cast<f32x4*>(Ab + krow * BM + 4 * c4) = ra[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_S; ++i) {
                int idx = tid + 256 * i;
                Ab[idx] = ras[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_V; ++i) {
            int krow = (tid >> 5) + 8 * i;
            *reinterpret_cast<f32x4*>(Bb + krow * BN + 4 * (tid & 31)) = rbv[i];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int acol = wm * WM * 32 + (lane & 31);
    const int bcol = wn * WN * 32 + (lane & 31);
    const int kh = lane >> 5;

    if (t_begin < t_end) {
        gload(t_begin);
        sstore(0);
        __syncthreads();
        for (int t = t_begin; t < t_end; ++t) {
            const int cur = (t - t_begin) & 1;
            const bool more = (t + 1) < t_end;
            if (more) gload(t + 1);
            const float* Ab = As + cur * BK * BM;
            const float* Bb = Bs + cur * BK * BN;
            // fragments double-buffered in registers: reads for k-pair kk+1 are in flight during the MFMAs of pair kk
            float a[2][WM], b[2][WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[0][i] = Ab[kh * BM + acol + 32 * i];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[0][j] = Bb[kh * BN + bcol + 32 * j];
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                if (kk + 1 < BK / 2) {
#pragma unroll
                    for (int i = 0; i < WM; ++i) a[(kk + 1) & 1][i] = Ab[(2 * kk + 2 + kh) * BM + acol + 32 * i];
#pragma unroll
                    for (int j = 0; j < WN; ++j) b[(kk + 1) & 1][j] = Bb[(2 * kk + 2 + kh) * BN + bcol + 32 * j];
                }
                __builtin_amdgcn_sched_barrier(0);     // keep the prefetch reads ABOVE this pair's MFMAs
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk & 1][i], b[kk & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) sstore(cur ^ 1);
            __syncthreads();
        }
    }

    float* out = p.out + (size_t)bz * p.Cout * p.Ktot;
    if ((p.Ktot & 3) == 0) {      // wide stores (see tile_rows4)
        float* scratch = smem + wave * (32 * 36);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                tile_rows4(scratch, acc[i][j], lane, [&](int rr, int cc, f32x4 v4) {
                    const int co = c0m + (wm * WM + i) * 32 + rr;
                    const int n = n0 + (wn * WN + j) * 32 + cc;
                    if (co < p.Cout && n < p.Ktot) *reinterpret_cast<f32x4*>(out + (size_t)co * p.Ktot + n) = v4;
                });
        return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int co = c0m + (wm * WM + i) * 32 + row;
            if (co >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = n0 + (wn * WN + j) * 32 + (lane & 31);
                if (n < p.Ktot) out[(size_t)co * p.Ktot + n] = acc[i][j][r];
            }
        }
}

// bf16-operand wgrad (BASELINE config 3).  Same GEMM view (M = Cout, N = (tap,c), K = pixels) but the MFMA wants 8
// consecutive k (pixels) per lane for a fixed row, while memory is pixel-major.  The transpose happens at staging: a
// thread takes TWO consecutive pixels x 4 channels (two 16-byte loads), rounds to bf16 and writes 4 dwords, each
// holding {pixel k, pixel k+1} of one channel, into k-contiguous rows [row][32 pixels + 8 pad].  Requires Cout % 4 == 0,
// C % 4 == 0 and row-segment addressing (Wo % 32 == 0).
template <int WAVES_M, int WAVES_N, int WM, int WN>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradParams p) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, BK = 32, LDH = 40;
    constexpr int A_IT = 16 * (BM / 4) / 256;     // (pixel pair, channel quad) items per thread
    constexpr int B_IT = 16 * (BN / 4) / 256;
    static_assert(WAVES_M * WAVES_N == 4 && BN == 128, "tile shape");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* Ah = reinterpret_cast<__bf16*>(smem);     // [2][BM][LDH]
    __bf16* Bh = Ah + 2 * BM * LDH;                   // [2][BN][LDH]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int work = xcd_remap(blockIdx.x, p.nx * p.ny * p.splits);
    const int bz = work / (p.nx * p.ny);
    const int bxy = work - bz * (p.nx * p.ny);
    const int n0 = (bxy % p.nx) * BN, c0m = (bxy / p.nx) * BM;
    const int t_begin = bz * p.steps_per_split;
    const int t_end = min(p.T, t_begin + p.steps_per_split);

    // B' items: quad = idx % 32 (4 columns of the N tile), pair = idx / 32 (pixels 2*pair, 2*pair+1 of the step)
    int bky[B_IT], bkx[B_IT], bc[B_IT], bpair[B_IT], bq[B_IT];
    bool bv[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int idx = tid + 256 * i;
        bq[i] = idx & 31;
        bpair[i] = idx >> 5;
        const int n = n0 + 4 * bq[i];
        bv[i] = n < p.Ktot;
        const int nn = bv[i] ? n : 0;
        const int tap = nn / p.C;
        bc[i] = nn - tap * p.C;
        bky[i] = tap / p.k - p.pad;
        bkx[i] = tap % p.k - p.pad;
    }
    int rs_ox0 = (t_begin * BK) % p.Wo, rs_oy = ((t_begin * BK) / p.Wo) % p.Ho, rs_b = (t_begin * BK) / p.Wo / p.Ho;

    f32x4 ra[A_IT][2], rb[B_IT][2];
    auto gload = [&](int t) {
        const int mbase = t * BK;
        constexpr int AQ = BM / 4;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + 256 * i;
            const int quad = idx % AQ, pair = idx / AQ;
            const int co = c0m + 4 * quad;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int m = mbase + 2 * pair + e;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (m < p.Mpix && co < p.Cout) v = *reinterpret_cast<const f32x4*>(p.dy + (size_t)m * p.Cout + co);
                ra[i][e] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int iy = rs_oy * p.stride + bky[i];
            const bool vy = bv[i] && iy >= 0 && iy < p.H && mbase < p.Mpix;
            const size_t rowoff = ((size_t)rs_b * p.H + (vy ? iy : 0)) * p.W;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ix = (rs_ox0 + 2 * bpair[i] + e) * p.stride + bkx[i];
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (vy && ix >= 0 && ix < p.W) {
                    const size_t pix = rowoff + ix;
                    v = *reinterpret_cast<const f32x4*>(p.x + pix * p.C + bc[i]);
                    if (p.amask) v *= p.amask[pix];
                }
                rb[i][e] = v;
            }
        }
        rs_ox0 += BK;
        if (rs_ox0 >= p.Wo) {
            rs_ox0 = 0;
            if (++rs_oy == p.Ho) { rs_oy = 0; ++rs_b; }
        }
    };
    auto pack2 = [](float lo, float hi) -> uint32_t {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 v = {lo, hi};
        bf16x2 h = __builtin_convertvector(v, bf16x2);
        return *reinterpret_cast<uint32_t*>(&h);
    };
    auto sstore = [&](int buf) {
        uint32_t* Ab = reinterpret_cast<uint32_t*>(Ah + buf * BM * LDH);
        uint32_t* Bb = reinterpret_cast<uint32_t*>(Bh + buf * BN * LDH);
        constexpr int AQ = BM / 4;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + 256 * i;
            const int quad = idx % AQ, pair = idx / AQ;
#pragma unroll
            for (int c = 0; c < 4; ++c) Ab[(4 * quad + c) * (LDH / 2) + pair] = pack2(ra[i][0][c], ra[i][1][c]);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) Bb[(4 * bq[i] + c) * (LDH / 2) + bpair[i]] = pack2(rb[i][0][c], rb[i][1][c]);
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int arow = wm * WM * 32 + (lane & 31), brow = wn * WN * 32 + (lane & 31);
    const int kh8 = 8 * (lane >> 5);

    if (t_begin < t_end) {
        gload(t_begin);
        sstore(0);
        __syncthreads();
        for (int t = t_begin; t < t_end; ++t) {
            const int cur = (t - t_begin) & 1;
            const bool more = (t + 1) < t_end;
            if (more) gload(t + 1);
            const __bf16* Ab = Ah + cur * BM * LDH;
            const __bf16* Bb = Bh + cur * BN * LDH;
#pragma unroll
            for (int gk = 0; gk < 2; ++gk) {
                bf16x8 ah[WM], bh[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) ah[i] = *reinterpret_cast<const bf16x8*>(Ab + (arow + 32 * i) * LDH + 16 * gk + kh8);
#pragma unroll
                for (int j = 0; j < WN; ++j) bh[j] = *reinterpret_cast<const bf16x8*>(Bb + (brow + 32 * j) * LDH + 16 * gk + kh8);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
            if (more) sstore(cur ^ 1);
            __syncthreads();
        }
    }
    float* out = p.out + (size_t)bz * p.Cout * p.Ktot;
    if ((p.Ktot & 3) == 0) {      // wide stores (see tile_rows4)
        float* scratch = smem + wave * (32 * 36);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                tile_rows4(scratch, acc[i][j], lane, [&](int rr, int cc, f32x4 v4) {
                    const int co = c0m + (wm * WM + i) * 32 + rr;
                    const int n = n0 + (wn * WN + j) * 32 + cc;
                    if (co < p.Cout && n < p.Ktot) *reinterpret_cast<f32x4*>(out + (size_t)co * p.Ktot + n) = v4;
                });
        return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int co = c0m + (wm * WM + i) * 32 + row;
            if (co >= p.Cout) continue;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = n0 + (wn * WN + j) * 32 + (lane & 31);
                if (n < p.Ktot) out[(size_t)co * p.Ktot + n] = acc[i][j][r];
            }
        }
}

template <int WAVES_M, int WAVES_N, int WM, int WN>
static int launch_wgrad_bf16_cfg(const WgradParams& p, hipStream_t s) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr size_t lds = (size_t)2 * (BM + BN) * 40 * 2;
    WgradParams pp = p;
    pp.rowseg = 1;
    pp.nx = cdiv(p.Ktot, BN);
    pp.ny = cdiv(p.Cout, BM);
    dim3 grid(pp.nx * pp.ny * p.splits);
    {
        const double by = 4.0 * ((double)p.B * p.H * p.W * p.C + (double)p.Mpix * p.Cout + (double)p.Cout * p.Ktot +
                                 (double)p.Mpix + (p.amask ? (double)p.B * p.H * p.W : 0.0));
        ProfScope ps(s, 3, 2.0 * p.Mpix * (double)p.Cout * p.Ktot, by, p.Cout, p.Ktot, p.Mpix, p.C, p.splits, 3500 + BM);
        hipLaunchKernelGGL((wgrad_bf16_kernel<WAVES_M, WAVES_N, WM, WN>), grid, dim3(256), lds, s, pp);
    }
    TG_CHECK_LAUNCH("wgrad_bf16_kernel");
    return TG_OK;
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                          size_t n, int splits) {
    if ((n & 3) == 0) {           // 16-byte accesses, fixed summation order over the slabs
        const size_t n4 = n >> 2;
        for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n4; idx += (size_t)gridDim.x * 256) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int z = 0; z < splits; ++z) v += reinterpret_cast<const f32x4*>(ws + (size_t)z * n)[idx];
            reinterpret_cast<f32x4*>(out)[idx] = v;
        }
        return;
    }
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += ws[(size_t)z * n + idx];
        out[idx] = v;
    }
}

template <int WAVES_M, int WAVES_N, int WM, int WN, bool SA, bool SB>
static int launch_wgrad_cfg(const WgradParams& p, hipStream_t s) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr size_t lds = (size_t)2 * 32 * (BM + BN) * sizeof(float);
    static LdsOptIn opt;
    auto kern = wgrad_kernel<WAVES_M, WAVES_N, WM, WN, SA, SB>;
    if (int rc = lds_opt_in(opt, reinterpret_cast<const void*>(kern), lds, "wgrad")) return rc;
    WgradParams pp = p;
    pp.rowseg = (p.Wo % 32 == 0 && !SB && !getenv("TG_NO_ROWSEG")) ? 1 : 0;
    pp.nx = cdiv(p.Ktot, BN);
    pp.ny = cdiv(p.Cout, BM);
    dim3 grid(pp.nx * pp.ny * p.splits);
    {
        const double by = 4.0 * ((double)p.B * p.H * p.W * p.C + (double)p.Mpix * p.Cout + (double)p.Cout * p.Ktot +
                                 (double)p.Mpix + (p.amask ? (double)p.B * p.H * p.W : 0.0));
        ProfScope ps(s, 1, 2.0 * p.Mpix * (double)p.Cout * p.Ktot, by, p.Cout, p.Ktot, p.Mpix, p.C, p.splits, BM);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, pp);
    }
    TG_CHECK_LAUNCH("wgrad_kernel");
    return TG_OK;
}

static int wgrad_bm(const TgConv* g) {
    if (g->Cout % 4 != 0 || g->Cout < 32) return 32;
    if (g->Cout >= 128 && g->Cin % 4 == 0) return 128;
    return 64;
}
static void wgrad_plan(const TgConv* g, int* splits, int* steps_per_split, int* T) {
    const int Mpix = g->B * g->Ho * g->Wo;
    const int Ktot = g->k * g->k * g->Cin;
    const int bm = wgrad_bm(g);
    const long tiles = (long)cdiv(g->Cout, bm) * cdiv(Ktot, 128);
    *T = cdiv(Mpix, 32);
    int smax = cdiv(*T, 4) < 512 ? cdiv(*T, 4) : 512;
    int sp = choose_splits(tiles, smax, 256 * (bm == 128 ? 2 : 3));
    *steps_per_split = cdiv(*T, sp);
    *splits = cdiv(*T, *steps_per_split);
}

int tg_colsum_launch(const float* x, int64_t rows, int C, float* out, float* ws, hipStream_t s);  // pointwise.hip
size_t tg_colsum_ws_floats(int64_t rows, int C);

extern "C" size_t tg_conv_wgrad_ws_bytes(const TgConv* g) {
    if (!g) return 0;
    int splits, sps, T;
    wgrad_plan(g, &splits, &sps, &T);
    size_t slabs = align_up((size_t)splits * g->Cout * g->k * g->k * g->Cin, 64);
    {
        size_t wf = align_up(wino_wgrad_ws_floats(g), 64);
        if (wf > slabs) slabs = wf;
        wf = align_up(wino22_wgrad_ws_floats(g), 64);
        if (wf > slabs) slabs = wf;
        wf = align_up(wino16_wgrad_ws_floats(g), 64);
        if (wf > slabs) slabs = wf;
    }
    if (s2d_ok(g)) {
        const TgConv g2 = s2d_geom(g);
        size_t wf = s2d_x_floats(g) + s2d_w_floats(g) + align_up(wino_wgrad_ws_floats(&g2), 64);
        if (wino_wgrad_ok(&g2, nullptr) && wf > slabs) slabs = wf;
        wf = s2d_x_floats(g) + s2d_w_floats(g) + align_up(wino16_wgrad_ws_floats(&g2), 64);
        if (wino16_wgrad_ok(&g2, nullptr) && wf > slabs) slabs = wf;
    }
    WgradParams sp = {};
    sp.C = g->Cin; sp.Cout = g->Cout; sp.k = g->k; sp.Mpix = g->B * g->Ho * g->Wo;
    if (smallconv_wgrad_applies(sp)) {
        size_t alt = align_up(smallconv_wgrad_ws_floats(sp), 64);
        if (alt > slabs) slabs = alt;
    }
    return (slabs + tg_colsum_ws_floats((int64_t)g->B * g->Ho * g->Wo, g->Cout)) * sizeof(float);
}

static int conv_wgrad_impl(const TgConv* g, const float* x, const float* in_mask, const float* dy, float* dw, float* db, float* ws,
                           size_t ws_bytes, tg_stream_t stream, const BnIn* in_bn);
extern "C" int tg_conv_wgrad(const TgConv* g, const float* x, const float* in_mask, const float* dy, float* dw,
                             float* db, float* ws, size_t ws_bytes, tg_stream_t stream) {
    return conv_wgrad_impl(g, x, in_mask, dy, dw, db, ws, ws_bytes, stream, nullptr);
}
extern "C" int tg_conv_wgrad_bnin(const TgConv* g, const float* x, const TgBnAct* bn, const float* dy, float* dw, float* db, float* ws,
                                  size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(bn && bn->mean && bn->rstd && bn->gamma && bn->beta, "tg_conv_wgrad_bnin: null pointer");
    TG_REQUIRE(tg_conv_bnin_supported(g, 1), "tg_conv_wgrad_bnin: geometry not supported (ask tg_conv_bnin_supported first)");
    const BnIn b = bnin_of(bn);
    return conv_wgrad_impl(g, x, nullptr, dy, dw, db, ws, ws_bytes, stream, &b);
}
static int conv_wgrad_impl(const TgConv* g, const float* x, const float* in_mask, const float* dy, float* dw, float* db, float* ws,
                           size_t ws_bytes, tg_stream_t stream, const BnIn* in_bn) {
    int rc = check_conv(g, "tg_conv_wgrad");
    if (rc) return rc;
    TG_REQUIRE(x && dy && dw && ws, "tg_conv_wgrad: null pointer");
    TG_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(ws), "tg_conv_wgrad: pointers must be 16-byte aligned");
    TG_REQUIRE(ws_bytes >= tg_conv_wgrad_ws_bytes(g), "tg_conv_wgrad: workspace too small (%zu < %zu)", ws_bytes,
               tg_conv_wgrad_ws_bytes(g));
    hipStream_t s = (hipStream_t)stream;
    WgradParams p = {};
    p.x = x; p.amask = in_mask; p.dy = dy;
    p.B = g->B; p.H = g->H; p.W = g->W; p.C = g->Cin; p.Ho = g->Ho; p.Wo = g->Wo; p.Cout = g->Cout;
    p.k = g->k; p.stride = g->stride; p.pad = g->pad;
    p.Mpix = g->B * g->Ho * g->Wo; p.Ktot = g->k * g->k * g->Cin;
    wgrad_plan(g, &p.splits, &p.steps_per_split, &p.T);
    if (in_bn) {
        p.in_bn = *in_bn;
        TG_REQUIRE(smallconv_wgrad_applies(p) && smallconv_bnin_wgrad_ok(p), "tg_conv_wgrad_bnin: geometry not supported");
    }
    if (smallconv_wgrad_applies(p)) {
        int db_done = 0;
        {
            const double by = 4.0 * ((double)p.B * p.H * p.W * p.C + (double)p.Mpix * p.Cout + (double)p.Cout * p.Ktot);
            ProfScope ps(s, 2, 2.0 * p.Mpix * (double)p.Cout * p.Ktot, by, p.Cout, p.Ktot, p.Mpix, p.C, 1, 2001);
            rc = smallconv_wgrad_launch(p, dw, ws, s, db, &db_done);
        }
        if (rc) return rc;
        if (db && !db_done) {
            size_t used = align_up(smallconv_wgrad_ws_floats(p), 64);
            rc = tg_colsum_launch(dy, (int64_t)p.Mpix, g->Cout, db, ws + used, s);
            if (rc) return rc;
        }
        return TG_OK;
    }
    if (s2d_ok(g)) {
        const TgConv g2 = s2d_geom(g);
        const bool w16 = wino16_wgrad_ok(&g2, nullptr);
        if (w16 || wino_wgrad_ok(&g2, nullptr)) {
            // 5x5 stride 2: Winograd wgrad over the (masked) space-to-depth input, then gather the 25 live taps
            float* x2 = ws;
            float* dw2 = ws + s2d_x_floats(g);
            float* wsr = dw2 + s2d_w_floats(g);
            hipLaunchKernelGGL(s2d_kernel, dim3(ew_grid((int64_t)g->B * g->H * g->W * (g->Cin / 4), 256)), dim3(256), 0, s, x, in_mask,
                               x2, g->B, g->H, g->W, g->Cin);
            TG_CHECK_LAUNCH("s2d_kernel");
            WgradParams p2 = p;
            p2.x = x2; p2.amask = nullptr; p2.H = g2.H; p2.W = g2.W; p2.C = g2.Cin; p2.k = 3; p2.stride = 1; p2.pad = 1;
            p2.Ktot = 9 * g2.Cin;
            {
                AlgScale sc(25.0 / 36.0);
                rc = w16 ? launch_wino16_wgrad(&g2, p2, dw2, wsr, s) : launch_wino_wgrad(&g2, p2, dw2, wsr, s);
            }
            if (rc) return rc;
            hipLaunchKernelGGL(w5x5_s2d_kernel, dim3(ew_grid((int64_t)g->Cout * 25 * g->Cin, 256)), dim3(256), 0, s, dw2, dw, g->Cout,
                               g->Cin, 0);
            TG_CHECK_LAUNCH("w5x5_s2d_kernel");
            if (db) {
                float* ws2 = wsr + align_up(w16 ? wino16_wgrad_ws_floats(&g2) : wino_wgrad_ws_floats(&g2), 64);
                rc = tg_colsum_launch(dy, (int64_t)p.Mpix, g->Cout, db, ws2, s);
                if (rc) return rc;
            }
            return TG_OK;
        }
    }
    if (wino16_wgrad_ok(g, in_mask)) {        // bf16 mode: the same with bf16 MFMA operands (wino16.inc)
        rc = launch_wino16_wgrad(g, p, dw, ws, s);
        if (rc) return rc;
        if (db) {
            float* ws2 = ws + align_up(wino16_wgrad_ws_floats(g), 64);
            rc = tg_colsum_launch(dy, (int64_t)p.Mpix, g->Cout, db, ws2, s);
            if (rc) return rc;
        }
        return TG_OK;
    }
    if (wino_wgrad_ok(g, in_mask)) {          // stride-1 3x3, 64-multiples of channels: Winograd F(3x3,2x2)
        rc = launch_wino_wgrad(g, p, dw, ws, s);
        if (rc) return rc;
        if (db) {
            float* ws2 = ws + align_up(wino_wgrad_ws_floats(g), 64);
            rc = tg_colsum_launch(dy, (int64_t)p.Mpix, g->Cout, db, ws2, s);
            if (rc) return rc;
        }
        return TG_OK;
    }
    if (wino22_wgrad_ok(g, in_mask)) {        // 4x4 stride 2, 64-multiples of channels: Winograd F(2x2,2x2)
        rc = launch_wino22_wgrad(g, p, dw, ws, s);
        if (rc) return rc;
        if (db) {
            float* ws2 = ws + align_up(wino22_wgrad_ws_floats(g), 64);
            rc = tg_colsum_launch(dy, (int64_t)p.Mpix, g->Cout, db, ws2, s);
            if (rc) return rc;
        }
        return TG_OK;
    }
    p.out = p.splits > 1 ? ws : dw;
    const int bm = wgrad_bm(g);
    const bool sb = (g->Cin % 4) != 0;
    const bool bf = g->precision == TG_PREC_BF16 && !sb && g->Cout % 4 == 0 && g->Wo % 32 == 0 && bm >= 64 && !getenv("TG_NO_BF16_WGRAD");
    if (bf && bm == 128) rc = launch_wgrad_bf16_cfg<2, 2, 2, 2>(p, s);
    else if (bf) rc = launch_wgrad_bf16_cfg<1, 4, 2, 1>(p, s);
    else if (bm == 128) rc = launch_wgrad_cfg<2, 2, 2, 2, false, false>(p, s);
    else if (bm == 64) rc = sb ? launch_wgrad_cfg<1, 4, 2, 1, false, true>(p, s) : launch_wgrad_cfg<1, 4, 2, 1, false, false>(p, s);
    else rc = sb ? launch_wgrad_cfg<1, 4, 1, 1, true, true>(p, s) : launch_wgrad_cfg<1, 4, 1, 1, true, false>(p, s);
    if (rc) return rc;
    const size_t n = (size_t)g->Cout * p.Ktot;
    if (p.splits > 1) {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(ew_grid((int64_t)((n & 3) ? n : n / 4), 256)), dim3(256), 0, s, ws, dw, n, p.splits);
        TG_CHECK_LAUNCH("slab_reduce_kernel");
    }
    if (db) {
        float* ws2 = ws + align_up((size_t)p.splits * n, 64);
        rc = tg_colsum_launch(dy, (int64_t)p.Mpix, g->Cout, db, ws2, s);
        if (rc) return rc;
    }
    return TG_OK;
}

__global__ __launch_bounds__(256) void fold_cin_kernel(const float* __restrict__ w3, float* __restrict__ w1, int n, int cin) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
        float v = 0.f;
        for (int c = 0; c < cin; ++c) v += w3[(size_t)idx * cin + c];
        w1[idx] = v;
    }
}
extern "C" int tg_fold_cin(const float* w3, int cout, int taps, int cin, float* w1, tg_stream_t stream) {
    TG_REQUIRE(w3 && w1 && cout > 0 && taps > 0 && cin > 0, "tg_fold_cin: bad arguments");
    int n = cout * taps;
    hipLaunchKernelGGL(fold_cin_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, w3, w1, n, cin);
    TG_CHECK_LAUNCH("fold_cin_kernel");
    return TG_OK;
}
