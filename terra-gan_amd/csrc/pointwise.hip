// Bandwidth-bound kernels of the train step: mask path, BatchNorm(+act) fwd/bwd, bilinear-up (+) concat,
// generator head, max-pool, loss reductions with their gradients, Adam.  All NHWC, fp32, coalesced along
// the channel (or pixel, for 1-channel images) axis; reductions are two-stage (per-block partials in a
// caller workspace, fixed-order final sum in fp64) so results are bitwise reproducible run to run.
#include <math.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";
void tg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* tg_last_error(void) { return g_err; }
extern "C" int tg_version(void) { return 100; }

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline hipStream_t S(tg_stream_t s) { return (hipStream_t)s; }

// =================================================================================================
// column reductions over a [rows][C] matrix (C <= 1024): thread = (row lane, column), partial[block][q][c]
// =================================================================================================
constexpr int CR_MAXPASS = 4;

struct ColGeom {
    int cpp;        // columns per pass = min(C, 256)
    int rlanes;     // row lanes = 256 / cpp
    int npass;      // ceil(C / cpp)
    int grid;       // blocks
    int64_t rows_per_block;
};
static ColGeom col_geom(int64_t rows, int C) {
    ColGeom g;
    g.cpp = C < 256 ? C : 256;
    g.rlanes = 256 / g.cpp;
    g.npass = cdiv(C, g.cpp);
    int64_t want = cdiv64(rows, (int64_t)g.rlanes * 16);
    if (want > 1024) want = 1024;
    if (want < 1) want = 1;
    g.rows_per_block = cdiv64(rows, want);
    g.grid = (int)cdiv64(rows, g.rows_per_block);
    return g;
}

template <int NQ, class F>
__global__ __launch_bounds__(256) void colreduce_kernel(F f, int64_t rows, int C, int cpp, int rlanes,
                                                        int64_t rows_per_block, float* __restrict__ partial) {
    __shared__ float red[NQ][256];
    const int tid = threadIdx.x;
    const int col = tid % cpp, rl = tid / cpp;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    float q[CR_MAXPASS][NQ];
#pragma unroll
    for (int ps = 0; ps < CR_MAXPASS; ++ps)
#pragma unroll
        for (int i = 0; i < NQ; ++i) q[ps][i] = 0.f;
    if (rl < rlanes) {
        for (int64_t r = r0 + rl; r < r1; r += rlanes) {
#pragma unroll
            for (int ps = 0; ps < CR_MAXPASS; ++ps) {
                int c = col + ps * cpp;
                if (c < C) f(r, c, q[ps]);
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < CR_MAXPASS; ++ps) {
        const int c = col + ps * cpp;
        if (ps * cpp >= C) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NQ; ++i) red[i][tid] = (rl < rlanes) ? q[ps][i] : 0.f;
        __syncthreads();
        if (rl == 0 && c < C) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                float s = 0.f;
                for (int l = 0; l < rlanes; ++l) s += red[i][l * cpp + col];
                partial[((size_t)blockIdx.x * NQ + i) * C + c] = s;
            }
        }
    }
}


// Vectorised first stage for C % 4 == 0: a thread owns 4 consecutive channels (one 16-byte load per operand and
// row), the block covers `qpp` channel quads x `rlanes` rows per pass.  Functors implement
//   void quad(int64_t row, int c0, float (&q)[NQ][4])   // accumulate channels c0..c0+3 of one row
// GROUPS > 1 (nblocks_g blocks per group): `rows` is the row count of ONE group; the tensor holds the groups back to back and
// each is reduced exactly as a launch of its own would reduce it (same block geometry, same order: the same bits) -- the
// stacked D(fake) / D(real) batch of the train step has BatchNorm statistics per pass (engine.discriminator_forward).
template <int NQ, class F>
__global__ __launch_bounds__(256) void colreduce4_kernel(F f0, int64_t rows, int C, int qpp, int rlanes, int64_t rows_per_block,
                                                         float* __restrict__ partial, int nblocks_g) {
    __shared__ float red[NQ][4][256];
    const int tid = threadIdx.x;
    const int cq = tid % qpp, rl = tid / qpp;
    const int grp = nblocks_g > 0 ? blockIdx.x / nblocks_g : 0;
    const F f = nblocks_g > 0 ? f0.group(grp, rows) : f0;
    const int64_t r0 = (int64_t)(blockIdx.x - grp * (nblocks_g > 0 ? nblocks_g : 0)) * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    const int nquads = C >> 2;
    for (int q0 = 0; q0 < nquads; q0 += qpp) {       // column passes (C > 1024 never happens: one pass for C <= 1024)
        const int c0 = (q0 + cq) * 4;
        float q[NQ][4];
#pragma unroll
        for (int i = 0; i < NQ; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) q[i][e] = 0.f;
        if (rl < rlanes && c0 < C) {
            // four rows per trip: their loads are independent of the running sums, so the compiler issues them together --
            // a rolled loop keeps one or two 16-byte loads per thread in flight, far too few bytes per CU for the HBM latency
#pragma unroll 4
            for (int64_t r = r0 + rl; r < r1; r += rlanes) f.quad(r, c0, q);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NQ; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[i][e][tid] = (rl < rlanes && c0 < C) ? q[i][e] : 0.f;
        __syncthreads();
        if (rl == 0 && c0 < C) {
#pragma unroll
            for (int i = 0; i < NQ; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sacc = 0.f;
                    for (int l = 0; l < rlanes; ++l) sacc += red[i][e][l * qpp + cq];
                    partial[((size_t)blockIdx.x * NQ + i) * C + c0 + e] = sacc;
                }
        }
    }
}
struct ColGeom4 {
    int qpp, rlanes, grid;
    int64_t rows_per_block;
};
static ColGeom4 col_geom4(int64_t rows, int C) {
    ColGeom4 g;
    const int nquads = C / 4;
    g.qpp = nquads < 256 ? nquads : 256;
    g.rlanes = 256 / g.qpp;
    static const int maxgrid = getenv("TG_COLRED_MAXGRID") ? atoi(getenv("TG_COLRED_MAXGRID")) : 512;
    int64_t want = cdiv64(rows, (int64_t)g.rlanes * 8);
    if (want > maxgrid) want = maxgrid;
    if (want < 1) want = 1;
    g.rows_per_block = cdiv64(rows, want);
    g.grid = (int)cdiv64(rows, g.rows_per_block);
    return g;
}

// Second stage shared by every column reduction: 16 columns x 16 partial-row lanes per 256-thread block;
// lane rl sums partial rows rl, rl+16, ... in fp64 (fixed order), the 16 lanes are combined through LDS in
// a fixed order as well -> deterministic.  Returns true in the thread that owns column *c_out (rl == 0).
constexpr int FR_COLS = 16, FR_LANES = 16;
template <int NQ>
__device__ __forceinline__ bool final_reduce(const float* __restrict__ partial, int nblocks, int C, double (&out)[NQ],
                                             int* c_out) {
    __shared__ double fr[NQ][FR_LANES][FR_COLS];
    const int col = threadIdx.x % FR_COLS, rl = threadIdx.x / FR_COLS;
    const int c = blockIdx.x * FR_COLS + col;
    double acc[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) acc[i] = 0.0;
    if (c < C) {
        // 4 partial rows per trip: the loads of a trip are independent, so their latencies overlap (a rolled loop with
        // the fp64 add in it exposed one L2 round trip per row and made this tiny kernel cost ~10 us)
        int b = rl;
        for (; b + 3 * FR_LANES < nblocks; b += 4 * FR_LANES) {
            float v[4][NQ];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < NQ; ++i) v[u][i] = partial[((size_t)(b + u * FR_LANES) * NQ + i) * C + c];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < NQ; ++i) acc[i] += (double)v[u][i];
        }
        for (; b < nblocks; b += FR_LANES)
#pragma unroll
            for (int i = 0; i < NQ; ++i) acc[i] += (double)partial[((size_t)b * NQ + i) * C + c];
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) fr[i][rl][col] = acc[i];
    __syncthreads();
    *c_out = c;
    if (rl != 0 || c >= C) return false;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        double s = 0.0;
        for (int l = 0; l < FR_LANES; ++l) s += fr[i][l][col];
        out[i] = s;
    }
    return true;
}
static inline int fr_grid(int C) { return cdiv(C, FR_COLS); }

// ---- plain column sum (conv bias gradient) --------------------------------------------------------
struct ColSumF {
    const float* x;
    int C;
    __device__ void operator()(int64_t r, int c, float (&q)[1]) const { q[0] += x[r * C + c]; }
};
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                           float* __restrict__ out) {
    double s[1];
    int c;
    if (final_reduce<1>(partial, nblocks, C, s, &c)) out[c] = (float)s[0];
}
size_t tg_colsum_ws_floats(int64_t rows, int C) {
    ColGeom g = col_geom(rows, C);
    return align_up((size_t)g.grid * C, 64);
}
int tg_colsum_launch(const float* x, int64_t rows, int C, float* out, float* ws, hipStream_t s) {
    TG_REQUIRE(C >= 1 && C <= 1024, "colsum: C=%d out of range [1,1024]", C);
    ColGeom g = col_geom(rows, C);
    ColSumF f{x, C};
    hipLaunchKernelGGL((colreduce_kernel<1, ColSumF>), dim3(g.grid), dim3(256), 0, s, f, rows, C, g.cpp, g.rlanes,
                       g.rows_per_block, ws);
    TG_CHECK_LAUNCH("colsum");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(fr_grid(C)), dim3(256), 0, s, ws, g.grid, C, out);
    TG_CHECK_LAUNCH("colsum_final");
    return TG_OK;
}

// =================================================================================================
// mask path
// =================================================================================================
__global__ __launch_bounds__(256) void mask_update_kernel(const float* __restrict__ mask, int B, int H, int W, int k,
                                                          int stride, int pad, int Ho, int Wo,
                                                          float* __restrict__ mask_out, float* __restrict__ ratio) {
    const int64_t total = (int64_t)B * Ho * Wo;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int ox = (int)(idx % Wo);
        int64_t t = idx / Wo;
        int oy = (int)(t % Ho);
        int b = (int)(t / Ho);
        float s = 0.f;
        for (int ky = 0; ky < k; ++ky) {
            int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= W) continue;
                s += mask[((int64_t)b * H + iy) * W + ix];
            }
        }
        const float on = s > 0.f ? 1.f : 0.f;
        mask_out[idx] = on;
        // pconv.py:39 evaluates `slide_winsize / (mask_sum + 1e-8)` as reciprocal(mask_sum + 1e-8) * slide_winsize
        // (Tensor.__rtruediv__): two roundings -- reproduced so that the ratio is bit-identical to the reference
        ratio[idx] = (1.0f / (s + 1e-8f)) * (float)(k * k) * on;
    }
}
extern "C" int tg_mask_update(const float* mask, int B, int H, int W, int k, int stride, int pad, int Ho, int Wo,
                              float* mask_out, float* ratio, tg_stream_t stream) {
    TG_REQUIRE(mask && mask_out && ratio, "tg_mask_update: null pointer");
    TG_REQUIRE(B > 0 && H > 0 && W > 0 && k > 0 && stride > 0 && pad >= 0, "tg_mask_update: bad dims");
    TG_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1 && Ho > 0 && Wo > 0,
               "tg_mask_update: Ho/Wo inconsistent");
    hipLaunchKernelGGL(mask_update_kernel, dim3(ew_grid((int64_t)B * Ho * Wo, 256)), dim3(256), 0, S(stream), mask, B, H, W,
                       k, stride, pad, Ho, Wo, mask_out, ratio);
    TG_CHECK_LAUNCH("mask_update_kernel");
    return TG_OK;
}

__host__ __device__ inline int floordiv2(int a) { return a >= 0 ? a / 2 : -((-a + 1) / 2); }

__global__ __launch_bounds__(256) void mask_up_merge_kernel(const float* __restrict__ up, const float* __restrict__ skip,
                                                            int B, int h, int w, int H, int W, int offy, int offx,
                                                            float* __restrict__ out) {
    const int64_t total = (int64_t)B * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int X = (int)(idx % W);
        int64_t t = idx / W;
        int Y = (int)(t % H);
        int b = (int)(t / H);
        int yu = Y - offy, xu = X - offx;
        float v = 0.f;
        if (yu >= 0 && yu < 2 * h && xu >= 0 && xu < 2 * w) v = up[((int64_t)b * h + (yu >> 1)) * w + (xu >> 1)];
        out[idx] = fmaxf(v, skip[idx]);
    }
}
extern "C" int tg_mask_up_merge(const float* up_mask, const float* skip_mask, int B, int h, int w, int H, int W,
                                float* out, tg_stream_t stream) {
    TG_REQUIRE(up_mask && skip_mask && out && B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "tg_mask_up_merge: bad arguments");
    hipLaunchKernelGGL(mask_up_merge_kernel, dim3(ew_grid((int64_t)B * H * W, 256)), dim3(256), 0, S(stream), up_mask,
                       skip_mask, B, h, w, H, W, floordiv2(H - 2 * h), floordiv2(W - 2 * w), out);
    TG_CHECK_LAUNCH("mask_up_merge_kernel");
    return TG_OK;
}


// ---- the whole mask pyramid of a generator forward in ONE launch ----------------------------------------------------
// pconv.py:33-40 x 14 layers + generator.py:51-54,68-74 x 7 decoder levels = 21 dependent element-wise ops over 1-channel
// maps that depend on the input mask only.  As separate launches they cost ~5 us each whatever their size (the 2x2 ... 16x16
// levels are pure launch latency).  Images are independent, so one 1024-thread workgroup walks all levels of ITS image with
// a workgroup barrier between levels (the levels' outputs are separate allocations, read only after the barrier that
// follows their last store).  The arithmetic per element is mask_update_kernel's / mask_up_merge_kernel's, bit for bit.
__device__ __forceinline__ void mask_update_one(const float* __restrict__ mask, int H, int W, int k, int stride, int pad, int Ho,
                                                int Wo, int b, int idx, float* __restrict__ mask_out, float* __restrict__ ratio) {
    const int ox = idx % Wo, oy = idx / Wo;
    float s = 0.f;
    if (k == 3) {
        // the pyramid's levels are all 3x3 (enc4-7, dec1-7): nine unconditional loads at clamped addresses, in flight together,
        // summed in the same order with zeros for the taps outside (s + 0 = s: the same bits) -- the rolled, branchy loop below
        // awaited one L2 round trip per tap, nine in a row per level of a 16-level dependent chain
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy * stride - pad + t / 3, ix = ox * stride - pad + t % 3;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float m = mask[((int64_t)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)];
            v[t] = ok ? m : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) s += v[t];
    } else {
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= W) continue;
                s += mask[((int64_t)b * H + iy) * W + ix];
            }
        }
    }
    const float on = s > 0.f ? 1.f : 0.f;
    const int64_t o = ((int64_t)b * Ho + oy) * Wo + ox;
    mask_out[o] = on;
    ratio[o] = (1.0f / (s + 1e-8f)) * (float)(k * k) * on;       // two roundings, as pconv.py:39 (see mask_update_kernel)
}
__global__ __launch_bounds__(1024) void mask_pyramid_kernel(const TgMaskPyramid pm) {
    const int b = blockIdx.x;
    for (int i = 0; i < pm.nops; ++i) {
        const TgMaskOp op = pm.op[i];
        if (op.kind == 0) {
            for (int idx = threadIdx.x; idx < op.Ho * op.Wo; idx += 1024)
                mask_update_one(op.in, op.H, op.W, op.k, op.stride, op.pad, op.Ho, op.Wo, b, idx, op.out, op.out2);
        } else {
            const int offy = floordiv2(op.Ho - 2 * op.H), offx = floordiv2(op.Wo - 2 * op.W);
            for (int idx = threadIdx.x; idx < op.Ho * op.Wo; idx += 1024) {
                const int X = idx % op.Wo, Y = idx / op.Wo;
                const int yu = Y - offy, xu = X - offx;
                float v = 0.f;
                if (yu >= 0 && yu < 2 * op.H && xu >= 0 && xu < 2 * op.W) v = op.in[((int64_t)b * op.H + (yu >> 1)) * op.W + (xu >> 1)];
                const int64_t o = ((int64_t)b * op.Ho + Y) * op.Wo + X;
                op.out[o] = fmaxf(v, op.in2[o]);
            }
        }
        __syncthreads();        // (drains this wave's stores, then the workgroup barrier: the next level reads them)
    }
}
extern "C" int tg_mask_pyramid(const TgMaskPyramid* pm, int B, tg_stream_t stream) {
    TG_REQUIRE(pm && B > 0 && pm->nops >= 1 && pm->nops <= TG_MASK_PYRAMID_MAX, "tg_mask_pyramid: bad arguments");
    for (int i = 0; i < pm->nops; ++i) {
        const TgMaskOp& op = pm->op[i];
        TG_REQUIRE(op.in && op.out && (op.kind == 0 ? op.out2 != nullptr : op.in2 != nullptr), "tg_mask_pyramid: op %d: null pointer", i);
        TG_REQUIRE(op.H > 0 && op.W > 0 && op.Ho > 0 && op.Wo > 0, "tg_mask_pyramid: op %d: bad dims", i);
        if (op.kind == 0)
            TG_REQUIRE(op.k > 0 && op.stride > 0 && op.pad >= 0 && op.Ho == (op.H + 2 * op.pad - op.k) / op.stride + 1 &&
                       op.Wo == (op.W + 2 * op.pad - op.k) / op.stride + 1, "tg_mask_pyramid: op %d: Ho/Wo inconsistent", i);
    }
    hipLaunchKernelGGL(mask_pyramid_kernel, dim3(B), dim3(1024), 0, S(stream), *pm);
    TG_CHECK_LAUNCH("mask_pyramid_kernel");
    return TG_OK;
}

// =================================================================================================
// BatchNorm
// =================================================================================================
// shifted sums: d = y - y[0][c]  ->  sum d, sum d^2 (no catastrophic cancellation when |mean| >> std)
struct BnStatF {
    const float* y;
    int C;
    __device__ BnStatF group(int g, int64_t rows_g) const { return BnStatF{y + (size_t)g * rows_g * C, C}; }
    __device__ void operator()(int64_t r, int c, float (&q)[2]) const {
        float d = y[r * C + c] - y[c];
        q[0] += d;
        q[1] += d * d;
    }
    __device__ void quad(int64_t r, int c0, float (&q)[2][4]) const {
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + r * C + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(y + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float d = v[e] - sh[e];
            q[0][e] += d;
            q[1][e] += d * d;
        }
    }
};
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                          const float* __restrict__ y, int64_t rows, float eps,
                                                          float momentum, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out, float* __restrict__ rm,
                                                          float* __restrict__ rv, int64_t* __restrict__ nbt) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    double q[2];
    int c;
    if (!final_reduce<2>(partial, nblocks, C, q, &c)) return;
    const double n = (double)rows;
    const double md = q[0] / n;
    double var = q[1] / n - md * md;
    if (var < 0.0) var = 0.0;
    const double mean = (double)y[c] + md;
    mean_out[c] = (float)mean;
    rstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rm) rm[c] = (float)((1.0 - (double)momentum) * (double)rm[c] + (double)momentum * mean);
    if (rv) rv[c] = (float)((1.0 - (double)momentum) * (double)rv[c] + (double)momentum * var * (n / (n - 1.0)));
}
extern "C" size_t tg_bn_ws_bytes(int64_t rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    ColGeom g = col_geom(rows, C);
    int grid = g.grid;
    if (C % 4 == 0) {
        ColGeom4 g4 = col_geom4(rows, C);
        if (g4.grid > grid) grid = g4.grid;
    }
    return align_up((size_t)grid * 5 * C, 64) * sizeof(float);
}
extern "C" int tg_bn_stats(const float* y, int64_t rows, int C, float eps, float momentum, float* save_mean,
                           float* save_rstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(y && save_mean && save_rstd && ws, "tg_bn_stats: null pointer");
    TG_REQUIRE(C >= 1 && C <= 1024, "tg_bn_stats: C=%d out of range [1,1024]", C);
    TG_REQUIRE(rows > 1, "tg_bn_stats: Expected more than 1 value per channel when training (rows=%lld)", (long long)rows);
    TG_REQUIRE(ws_bytes >= tg_bn_ws_bytes(rows, C), "tg_bn_stats: workspace too small");
    BnStatF f{y, C};
    int nblocks;
    if (C % 4 == 0) {
        ColGeom4 g = col_geom4(rows, C);
        nblocks = g.grid;
        hipLaunchKernelGGL((colreduce4_kernel<2, BnStatF>), dim3(g.grid), dim3(256), 0, S(stream), f, rows, C, g.qpp, g.rlanes,
                           g.rows_per_block, ws, 0);
    } else {
        ColGeom g = col_geom(rows, C);
        nblocks = g.grid;
        hipLaunchKernelGGL((colreduce_kernel<2, BnStatF>), dim3(g.grid), dim3(256), 0, S(stream), f, rows, C, g.cpp, g.rlanes,
                           g.rows_per_block, ws);
    }
    TG_CHECK_LAUNCH("bn_stats");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(fr_grid(C)), dim3(256), 0, S(stream), ws, nblocks, C, y, rows, eps,
                       momentum, save_mean, save_rstd, running_mean, running_var, num_batches_tracked);
    TG_CHECK_LAUNCH("bn_finalize");
    return TG_OK;
}

__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* rm, const float* rv, int C, float eps, float* mean,
                                                            float* rstd) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    mean[c] = rm[c];
    rstd[c] = 1.0f / sqrtf(rv[c] + eps);
}
extern "C" int tg_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                                float* rstd, tg_stream_t stream) {
    TG_REQUIRE(running_mean && running_var && mean && rstd && C > 0, "tg_bn_eval_stats: bad arguments");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S(stream), running_mean, running_var, C, eps,
                       mean, rstd);
    TG_CHECK_LAUNCH("bn_eval_stats_kernel");
    return TG_OK;
}

// C % 4 == 0: a thread owns one channel quad (its per-channel constants live in registers) and walks rows, so the
// inner loop is two 16-byte accesses and 4 FMAs -- no integer division.  qpp = min(C/4, 256) quads per pass.
__global__ __launch_bounds__(256) void bn_act_fwd4_kernel(const float* __restrict__ y, int64_t rows, int C, int qpp, int rlanes,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int act, float slope, float* __restrict__ out) {
    const int cq = threadIdx.x % qpp, rl = threadIdx.x / qpp;
    if (rl >= rlanes) return;
    const int nquads = C >> 2;
    for (int q0 = blockIdx.y * qpp; q0 < nquads; q0 += gridDim.y * qpp) {
        const int c0 = (q0 + cq) * 4;
        if (c0 >= C) continue;
        const f32x4 mv = *reinterpret_cast<const f32x4*>(mean + c0), rv = *reinterpret_cast<const f32x4*>(rstd + c0);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + c0), bv = *reinterpret_cast<const f32x4*>(beta + c0);
        for (int64_t r = (int64_t)blockIdx.x * rlanes + rl; r < rows; r += (int64_t)gridDim.x * rlanes) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(y + r * C + c0);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = apply_act(bn_affine(v[e], mv[e], rv[e], gv[e], bv[e]), act, slope);
            *reinterpret_cast<f32x4*>(out + r * C + c0) = o;
        }
    }
}
__global__ __launch_bounds__(256) void bn_act_fwd1_kernel(const float* __restrict__ y, int64_t total, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int act, float slope, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float xh = (y[i] - mean[c]) * rstd[c];
        out[i] = apply_act(xh * gamma[c] + beta[c], act, slope);
    }
}
struct RowGeom {
    int qpp, rlanes;
    dim3 grid;
};
static RowGeom row_geom(int64_t rows, int C) {
    RowGeom g;
    const int nquads = C / 4;
    g.qpp = nquads < 256 ? nquads : 256;
    g.rlanes = 256 / g.qpp;
    int64_t gx = cdiv64(rows, (int64_t)g.rlanes * 4);
    if (gx > 2048) gx = 2048;
    if (gx < 1) gx = 1;
    g.grid = dim3((unsigned)gx, (unsigned)cdiv(nquads, g.qpp));
    return g;
}
extern "C" int tg_bn_act_fwd(const float* y, int64_t rows, int C, const float* mean, const float* rstd, const float* gamma,
                             const float* beta, int act, float slope, float* out, tg_stream_t stream) {
    TG_REQUIRE(y && mean && rstd && gamma && beta && out && rows > 0 && C > 0, "tg_bn_act_fwd: bad arguments");
    if (C % 4 == 0) {
        RowGeom g = row_geom(rows, C);
        hipLaunchKernelGGL(bn_act_fwd4_kernel, g.grid, dim3(256), 0, S(stream), y, rows, C, g.qpp, g.rlanes, mean, rstd, gamma,
                           beta, act, slope, out);
    } else {
        hipLaunchKernelGGL(bn_act_fwd1_kernel, dim3(ew_grid(rows * C, 256)), dim3(256), 0, S(stream), y, rows * C, C, mean, rstd,
                           gamma, beta, act, slope, out);
    }
    TG_CHECK_LAUNCH("bn_act_fwd_kernel");
    return TG_OK;
}

__device__ __forceinline__ float act_grad(float z, int act, float slope) {
    if (act == TG_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == TG_ACT_LEAKY) return z > 0.f ? 1.f : slope;
    return 1.f;
}

// ---- BatchNorm of SMALL [rows][C] maps: statistics, finalisation and apply in ONE launch ------------------------------------
// Channels are independent, so a workgroup that owns 16 channels (4 quads x 64 row lanes) needs nobody else: it sums its
// columns (the same shifted sums as BnStatF, fp32 per lane, combined in fp64 in a fixed order), finishes mean / rstd / the
// running statistics, and applies scale + shift + activation on a second sweep over rows that are still in its L2.  Used
// when there are few rows (the 2x2 ... 16x16 levels of the U-Net: rows <= BN_SMALL_ROWS): there the three-launch form is pure
// launch latency (3 x ~5 us for microseconds of work); large maps keep the bandwidth-shaped kernels.
constexpr int BN_SMALL_ROWS = 2048;
static bool bn_small_ok(int64_t rows, int C) {
    static const bool off = getenv("TG_NO_BN_SMALL") != nullptr;
    return !off && rows <= BN_SMALL_ROWS && C % 16 == 0 && C >= 64;
}
template <int NQ>
__device__ __forceinline__ void quad_lane_reduce(const float (&q)[NQ][4], double (&out)[NQ][4], double (*red)[4][64][4], int cq, int rl) {
    // red[NQ][4 quads][64 lanes][4]: thread t < NQ*16 sums the 64 row lanes of ONE (sum, quad, channel) in a fixed order
    // (fp64) and parks the result in lane slot 0; every thread of the quad then reads its NQ x 4 totals
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[i][cq][rl][e] = (double)q[i][e];
    __syncthreads();
    const int t = threadIdx.x;
    double s = 0.0;
    if (t < NQ * 16) {
        const int i = t >> 4, qd = (t >> 2) & 3, e = t & 3;
        for (int l = 0; l < 64; ++l) s += red[i][qd][l][e];
    }
    __syncthreads();
    if (t < NQ * 16) red[t >> 4][(t >> 2) & 3][0][t & 3] = s;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) out[i][e] = red[i][cq][0][e];
}
__global__ __launch_bounds__(256) void bn_fwd_small_kernel(const float* __restrict__ y, int rows, int C, float eps, float momentum,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                           float slope, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                           float* __restrict__ rm, float* __restrict__ rv, int64_t* __restrict__ nbt,
                                                           float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double bn_small_red[];
    double (*red)[4][64][4] = reinterpret_cast<double (*)[4][64][4]>(bn_small_red);
    const int cq = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int c0 = blockIdx.x * 16 + 4 * cq;
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    const f32x4 sh = *reinterpret_cast<const f32x4*>(y + c0);
    float q[2][4] = {};
#pragma unroll 4
    for (int r = rl; r < rows; r += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + (size_t)r * C + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = v[e] - sh[e];
            q[0][e] += d;
            q[1][e] += d * d;
        }
    }
    double sq[2][4];
    quad_lane_reduce<2>(q, sq, red, cq, rl);
    const double n = (double)rows;
    f32x4 mv, rsv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double md = sq[0][e] / n;
        double var = sq[1][e] / n - md * md;
        if (var < 0.0) var = 0.0;
        const double mean = (double)sh[e] + md;
        mv[e] = (float)mean;
        rsv[e] = (float)(1.0 / sqrt(var + (double)eps));
        if (rl == 0) {
            if (rm) rm[c0 + e] = (float)((1.0 - (double)momentum) * (double)rm[c0 + e] + (double)momentum * mean);
            if (rv) rv[c0 + e] = (float)((1.0 - (double)momentum) * (double)rv[c0 + e] + (double)momentum * var * (n / (n - 1.0)));
        }
    }
    if (rl == 0) {
        *reinterpret_cast<f32x4*>(mean_out + c0) = mv;
        *reinterpret_cast<f32x4*>(rstd_out + c0) = rsv;
    }
    if (!out) return;
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + c0), bv = *reinterpret_cast<const f32x4*>(beta + c0);
#pragma unroll 4
    for (int r = rl; r < rows; r += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + (size_t)r * C + c0);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = apply_act((v[e] - mv[e]) * rsv[e] * gv[e] + bv[e], act, slope);
        *reinterpret_cast<f32x4*>(out + (size_t)r * C + c0) = o;
    }
}
// training-mode BatchNorm forward: batch statistics + running update (tg_bn_stats) and, when `out` != NULL, the affine +
// activation apply (tg_bn_act_fwd) -- one launch for small maps, the two calls' three launches otherwise
extern "C" int tg_bn_fwd(const float* y, int64_t rows, int C, float eps, float momentum, const float* gamma, const float* beta,
                         int act, float slope, float* save_mean, float* save_rstd, float* running_mean, float* running_var,
                         int64_t* num_batches_tracked, float* out, float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(y && save_mean && save_rstd && (!out || (gamma && beta)), "tg_bn_fwd: null pointer");
    TG_REQUIRE(rows > 1, "tg_bn_fwd: Expected more than 1 value per channel when training (rows=%lld)", (long long)rows);
    if (bn_small_ok(rows, C)) {
        hipLaunchKernelGGL(bn_fwd_small_kernel, dim3(C / 16), dim3(256), 2 * 4 * 64 * 4 * sizeof(double), S(stream), y, (int)rows, C, eps,
                           momentum, gamma, beta, act, slope, save_mean, save_rstd, running_mean, running_var, num_batches_tracked, out);
        TG_CHECK_LAUNCH("bn_fwd_small_kernel");
        return TG_OK;
    }
    if (int rc = tg_bn_stats(y, rows, C, eps, momentum, save_mean, save_rstd, running_mean, running_var, num_batches_tracked, ws,
                             ws_bytes, stream)) return rc;
    if (!out) return TG_OK;
    return tg_bn_act_fwd(y, rows, C, save_mean, save_rstd, gamma, beta, act, slope, out, stream);
}

__device__ __forceinline__ float act_grad(float z, int act, float slope);
__global__ __launch_bounds__(256) void bn_bwd_small_kernel(const float* __restrict__ dout, const float* __restrict__ y, int rows, int C,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                           float slope, const float* __restrict__ ratio, float* __restrict__ dy,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias) {
    extern __shared__ __attribute__((aligned(16))) double bn_small_red[];
    double (*red)[4][64][4] = reinterpret_cast<double (*)[4][64][4]>(bn_small_red);
    const int cq = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int c0 = blockIdx.x * 16 + 4 * cq;
    const f32x4 mv = *reinterpret_cast<const f32x4*>(mean + c0), rv = *reinterpret_cast<const f32x4*>(rstd + c0);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + c0), bv = *reinterpret_cast<const f32x4*>(beta + c0);
    float q[5][4] = {};
#pragma unroll 2
    for (int r = rl; r < rows; r += 64) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + (size_t)r * C + c0);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dout + (size_t)r * C + c0);
        const float rr = ratio ? ratio[r] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (yv[e] - mv[e]) * rv[e];
            const float g = dv[e] * act_grad(xh * gv[e] + bv[e], act, slope);
            q[0][e] += g;
            q[1][e] += g * xh;
            q[2][e] += rr * g;
            q[3][e] += rr * xh;
            q[4][e] += rr;
        }
    }
    double sq[5][4];
    quad_lane_reduce<5>(q, sq, red, cq, rl);
    const double n = (double)rows;
    f32x4 dgv, dbv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        dbv[e] = (float)sq[0][e];
        dgv[e] = (float)sq[1][e];
        if (rl == 0 && dbias)
            dbias[c0 + e] = (float)((double)gv[e] * (double)rv[e] * (sq[2][e] - sq[0][e] / n * sq[4][e] - sq[1][e] / n * sq[3][e]));
    }
    const float inv_n = 1.0f / (float)rows;
    // dy may alias dout (in place): every thread re-reads exactly the elements it overwrites, after all sums are complete
#pragma unroll 2
    for (int r = rl; r < rows; r += 64) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + (size_t)r * C + c0);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dout + (size_t)r * C + c0);
        const float rr = ratio ? ratio[r] : 1.f;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (yv[e] - mv[e]) * rv[e];
            const float g = dv[e] * act_grad(xh * gv[e] + bv[e], act, slope);
            o[e] = gv[e] * rv[e] * (g - dbv[e] * inv_n - xh * dgv[e] * inv_n) * rr;
        }
        *reinterpret_cast<f32x4*>(dy + (size_t)r * C + c0) = o;
    }
    if (rl == 0) {
        *reinterpret_cast<f32x4*>(dgamma + c0) = dgv;
        *reinterpret_cast<f32x4*>(dbeta + c0) = dbv;
    }
}

struct BnBwdF {
    const float* dout;
    const float* y;
    const float* mean;
    const float* rstd;
    const float* gamma;
    const float* beta;
    int C, act;
    float slope;
    const float* ratio;   // optional per-row scale of the conv output (partial conv)
    __device__ BnBwdF group(int g, int64_t rows_g) const {          // per-group statistics live at mean / rstd + g * C
        const size_t o = (size_t)g * rows_g * C;
        return BnBwdF{dout + o, y + o, mean + (size_t)g * C, rstd + (size_t)g * C, gamma, beta, C, act, slope,
                      ratio ? ratio + (size_t)g * rows_g : nullptr};
    }
    __device__ void operator()(int64_t r, int c, float (&q)[5]) const {
        float xh = (y[r * C + c] - mean[c]) * rstd[c];
        float g = dout[r * C + c] * act_grad(xh * gamma[c] + beta[c], act, slope);
        float rr = ratio ? ratio[r] : 1.f;
        q[0] += g;
        q[1] += g * xh;
        q[2] += rr * g;      // the three extra sums give the conv-bias gradient sum_rows ratio*dy in closed form
        q[3] += rr * xh;
        q[4] += rr;
    }
    __device__ void quad(int64_t r, int c0, float (&q)[5][4]) const {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(y + r * C + c0);
        const f32x4 dv = *reinterpret_cast<const f32x4*>(dout + r * C + c0);
        const f32x4 mv = *reinterpret_cast<const f32x4*>(mean + c0), rv = *reinterpret_cast<const f32x4*>(rstd + c0);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + c0), bv = *reinterpret_cast<const f32x4*>(beta + c0);
        const float rr = ratio ? ratio[r] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float xh = (yv[e] - mv[e]) * rv[e];
            float g = dv[e] * act_grad(xh * gv[e] + bv[e], act, slope);
            q[0][e] += g;
            q[1][e] += g * xh;
            q[2][e] += rr * g;
            q[3][e] += rr * xh;
            q[4][e] += rr;
        }
    }
};
__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* __restrict__ partial, int nblocks, int C, double n,
                                                           const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ dbias) {
    double q[5];
    int c;
    if (!final_reduce<5>(partial, nblocks, C, q, &c)) return;
    dbeta[c] = (float)q[0];
    dgamma[c] = (float)q[1];
    // dy = gamma*rstd*(g - dbeta/n - xhat*dgamma/n)  =>  sum_rows ratio*dy
    if (dbias) dbias[c] = (float)((double)gamma[c] * (double)rstd[c] * (q[2] - q[0] / n * q[4] - q[1] / n * q[3]));
}
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                            int64_t rows, int C, int qpp, int rlanes,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int act, float slope, const float* __restrict__ ratio,
                                                            const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                            float* __restrict__ dy) {
    const int cq = threadIdx.x % qpp, rl = threadIdx.x / qpp;
    if (rl >= rlanes) return;
    const float inv_n = 1.0f / (float)rows;
    const int nquads = C >> 2;
    for (int q0 = blockIdx.y * qpp; q0 < nquads; q0 += gridDim.y * qpp) {
        const int c0 = (q0 + cq) * 4;
        if (c0 >= C) continue;
        const f32x4 mv = *reinterpret_cast<const f32x4*>(mean + c0), rv = *reinterpret_cast<const f32x4*>(rstd + c0);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + c0), bv = *reinterpret_cast<const f32x4*>(beta + c0);
        const f32x4 dgv = *reinterpret_cast<const f32x4*>(dgamma + c0), dbv = *reinterpret_cast<const f32x4*>(dbeta + c0);
        for (int64_t r = (int64_t)blockIdx.x * rlanes + rl; r < rows; r += (int64_t)gridDim.x * rlanes) {
            const f32x4 yv = *reinterpret_cast<const f32x4*>(y + r * C + c0);
            const f32x4 dv = *reinterpret_cast<const f32x4*>(dout + r * C + c0);
            const float rr = ratio ? ratio[r] : 1.f;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xh = (yv[e] - mv[e]) * rv[e];
                float g = dv[e] * act_grad(xh * gv[e] + bv[e], act, slope);
                o[e] = gv[e] * rv[e] * (g - dbv[e] * inv_n - xh * dgv[e] * inv_n) * rr;
            }
            *reinterpret_cast<f32x4*>(dy + r * C + c0) = o;
        }
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply1_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                            int64_t rows, int C, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int act, float slope,
                                                            const float* __restrict__ ratio, const float* __restrict__ dgamma,
                                                            const float* __restrict__ dbeta, float* __restrict__ dy) {
    const int64_t total = rows * C;
    const float inv_n = 1.0f / (float)rows;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t r = i / C;
        float xh = (y[i] - mean[c]) * rstd[c];
        float g = dout[i] * act_grad(xh * gamma[c] + beta[c], act, slope);
        float v = gamma[c] * rstd[c] * (g - dbeta[c] * inv_n - xh * dgamma[c] * inv_n);
        if (ratio) v *= ratio[r];
        dy[i] = v;
    }
}
// ---- BatchNorm backward whose incoming gradient is the input gradient of a C -> 1 channel 3x3 / stride 1 / pad 1 convolution ----
// (dec1 under `final`, generator.py:29,56): dout[b][y][x][c] = sum_{ky,kx} dz[b][y+1-ky][x+1-kx] * w[ky][kx][c] costs nine FMAs per
// element from a 1-channel tensor, so it is RECOMPUTED in both passes (sums, apply) instead of being written once by the dgrad
// kernel and read twice: per step one 268 MB write and two 268 MB reads less at the headline size.  Same block geometry and
// partial layout as colreduce4_kernel<5, BnBwdF> (the shared finaliser follows); a block walks a contiguous row range, so the
// pixel coordinates advance incrementally (one division per thread).
struct BnConv1 {
    const float* dz;     // [B][H][W]
    const float* w;      // [3][3][C]
    int H, W;
};
struct Conv1Walk {
    int b, y, x;
    __device__ void init(int64_t r, int H, int W) {
        x = (int)(r % W);
        const int64_t t = r / W;
        y = (int)(t % H);
        b = (int)(t / H);
    }
    __device__ void advance(int step, int H, int W) {
        x += step;
        while (x >= W) {
            x -= W;
            if (++y == H) { y = 0; ++b; }
        }
    }
};
// SHARE form (qpp = 16 / 32 / 64 channel quads: the threads of a pixel are an aligned lane group of that width): lane t < 9 of the
// group fetches tap t -- one address, one bounds test, one load per thread and row instead of nine of each, all sixteen-fold
// redundant -- and the nine words go round by ds_bpermute (`__shfl` inside the group).  The vector ALU, not HBM, bounded the
// first form: ~180 instructions per thread and row, half of them tap addressing.
__device__ __forceinline__ f32x4 conv1_dgrad_quad_shared(const BnConv1& cv, const Conv1Walk& p, const f32x4 (&wq)[9], int ky, int kx, int width) {
    const int yy = p.y + 1 - ky, xx = p.x + 1 - kx;
    const bool ok = yy >= 0 && yy < cv.H && xx >= 0 && xx < cv.W;
    const float d = cv.dz[(size_t)p.b * cv.H * cv.W + (ok ? yy * cv.W + xx : 0)];
    const float mine = ok ? d : 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float dd = __shfl(mine, t, width);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(dd, wq[t][e], acc[e]);
    }
    return acc;
}
__device__ __forceinline__ f32x4 conv1_dgrad_quad(const BnConv1& cv, const Conv1Walk& p, const f32x4 (&wq)[9]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* img = cv.dz + (size_t)p.b * cv.H * cv.W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = p.y + 1 - ky;
        const bool vy = yy >= 0 && yy < cv.H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = p.x + 1 - kx;
            const bool ok = vy && xx >= 0 && xx < cv.W;
            const float d = img[ok ? yy * cv.W + xx : 0];            // unconditional load at a clamped address
            const float dd = ok ? d : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaf(dd, wq[ky * 3 + kx][e], acc[e]);
        }
    }
    return acc;
}
template <bool SHARE>
__global__ __launch_bounds__(256) void bn_bwd_conv1_reduce_kernel(const BnBwdF f, const BnConv1 cv, int64_t rows, int C, int qpp, int rlanes,
                                                                  int64_t rows_per_block, float* __restrict__ partial) {
    __shared__ float red[5][4][256];
    const int tid = threadIdx.x;
    const int cq = tid % qpp, rl = tid / qpp;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    const int c0 = cq * 4;                                            // (one column pass: C <= 1024)
    float q[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) q[i][e] = 0.f;
    const bool on = rl < rlanes && c0 < C;
    if (on) {
        f32x4 wq[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wq[t] = *reinterpret_cast<const f32x4*>(cv.w + (size_t)t * C + c0);
        const f32x4 mv = *reinterpret_cast<const f32x4*>(f.mean + c0), rv = *reinterpret_cast<const f32x4*>(f.rstd + c0);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(f.gamma + c0), bv = *reinterpret_cast<const f32x4*>(f.beta + c0);
        Conv1Walk p;
        p.init(r0 + rl, cv.H, cv.W);
        const int my_t = cq < 9 ? cq : 0, my_ky = my_t / 3, my_kx = my_t - 3 * my_ky;
#pragma unroll 4
        for (int64_t r = r0 + rl; r < r1; r += rlanes) {
            const f32x4 yv = *reinterpret_cast<const f32x4*>(f.y + r * C + c0);
            const f32x4 dv = SHARE ? conv1_dgrad_quad_shared(cv, p, wq, my_ky, my_kx, qpp) : conv1_dgrad_quad(cv, p, wq);
            const float rr = f.ratio ? f.ratio[r] : 1.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (yv[e] - mv[e]) * rv[e];
                const float g = dv[e] * act_grad(xh * gv[e] + bv[e], f.act, f.slope);
                q[0][e] += g;
                q[1][e] += g * xh;
                q[2][e] += rr * g;
                q[3][e] += rr * xh;
                q[4][e] += rr;
            }
            p.advance(rlanes, cv.H, cv.W);
        }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[i][e][tid] = on ? q[i][e] : 0.f;
    __syncthreads();
    if (rl == 0 && c0 < C) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float sacc = 0.f;
                for (int l = 0; l < rlanes; ++l) sacc += red[i][e][l * qpp + cq];
                partial[((size_t)blockIdx.x * 5 + i) * C + c0 + e] = sacc;
            }
    }
}
template <bool SHARE>
__global__ __launch_bounds__(256) void bn_bwd_conv1_apply_kernel(const BnBwdF f, const BnConv1 cv, int64_t rows, int C, int qpp, int rlanes,
                                                                 int64_t rows_per_block, const float* __restrict__ dgamma,
                                                                 const float* __restrict__ dbeta, float* __restrict__ dy) {
    const int cq = threadIdx.x % qpp, rl = threadIdx.x / qpp;
    const int c0 = cq * 4;
    if (rl >= rlanes || c0 >= C) return;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    const float inv_n = 1.0f / (float)rows;
    f32x4 wq[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wq[t] = *reinterpret_cast<const f32x4*>(cv.w + (size_t)t * C + c0);
    const f32x4 mv = *reinterpret_cast<const f32x4*>(f.mean + c0), rv = *reinterpret_cast<const f32x4*>(f.rstd + c0);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(f.gamma + c0), bv = *reinterpret_cast<const f32x4*>(f.beta + c0);
    const f32x4 dgv = *reinterpret_cast<const f32x4*>(dgamma + c0), dbv = *reinterpret_cast<const f32x4*>(dbeta + c0);
    Conv1Walk p;
    p.init(r0 + rl, cv.H, cv.W);
    const int my_t = cq < 9 ? cq : 0, my_ky = my_t / 3, my_kx = my_t - 3 * my_ky;
#pragma unroll 2
    for (int64_t r = r0 + rl; r < r1; r += rlanes) {
        const f32x4 yv = *reinterpret_cast<const f32x4*>(f.y + r * C + c0);
        const f32x4 dv = SHARE ? conv1_dgrad_quad_shared(cv, p, wq, my_ky, my_kx, qpp) : conv1_dgrad_quad(cv, p, wq);
        const float rr = f.ratio ? f.ratio[r] : 1.f;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (yv[e] - mv[e]) * rv[e];
            const float g = dv[e] * act_grad(xh * gv[e] + bv[e], f.act, f.slope);
            o[e] = gv[e] * rv[e] * (g - dbv[e] * inv_n - xh * dgv[e] * inv_n) * rr;
        }
        *reinterpret_cast<f32x4*>(dy + r * C + c0) = o;
        p.advance(rlanes, cv.H, cv.W);
    }
}
extern "C" size_t tg_bn_conv1_ws_bytes(int64_t rows, int C) {
    const size_t a = tg_bn_ws_bytes(rows, C), b = align_up((size_t)4096 * 5 * C, 64) * sizeof(float);      // up to 4096 first-stage blocks
    return a > b ? a : b;
}
extern "C" int tg_bn_bwd_conv1_supported(int64_t rows, int C) {
    return C % 4 == 0 && C >= 4 && C <= 1024 && rows > 0 && !bn_small_ok(rows, C) && !getenv("TG_NO_BN_CONV1") ? 1 : 0;
}
extern "C" int tg_bn_act_bwd_conv1(const float* dz, const float* w, int B, int H, int W, const float* y, int C, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, int act, float slope, const float* ratio,
                                   float* dy, float* dgamma, float* dbeta, float* dbias, float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(dz && w && y && mean && rstd && gamma && beta && dy && dgamma && dbeta && ws && B > 0 && H > 0 && W > 0,
               "tg_bn_act_bwd_conv1: bad arguments");
    const int64_t rows = (int64_t)B * H * W;
    TG_REQUIRE(tg_bn_bwd_conv1_supported(rows, C), "tg_bn_act_bwd_conv1: geometry not supported (ask tg_bn_bwd_conv1_supported first)");
    TG_REQUIRE(ws_bytes >= tg_bn_conv1_ws_bytes(rows, C), "tg_bn_act_bwd_conv1: workspace too small");
    const BnBwdF f{nullptr, y, mean, rstd, gamma, beta, C, act, slope, ratio};
    const BnConv1 cv{dz, w, H, W};
    ColGeom4 g = col_geom4(rows, C);
    {
        // more, shorter blocks than the plain reduction takes: this first stage is arithmetic (nine FMAs + the BatchNorm algebra per
        // element) behind dependent loads, two waves per SIMD cannot hide them
        static const int cgrid_env = getenv("TG_BN_CONV1_GRID") ? atoi(getenv("TG_BN_CONV1_GRID")) : 2048;
        const int cgrid = cgrid_env < 1 ? 1 : (cgrid_env > 4096 ? 4096 : cgrid_env);
        int64_t want = cdiv64(rows, (int64_t)g.rlanes * 8);
        if (want > cgrid) want = cgrid;
        if (want < 1) want = 1;
        g.rows_per_block = cdiv64(rows, want);
        g.grid = (int)cdiv64(rows, g.rows_per_block);
    }
    // lane-group sharing of the nine taps: the threads of a pixel must be a whole, aligned power-of-two set of lanes of ONE wave
    // (qpp = 16 / 32 / 64); they share the row index, hence every loop trip and the early exits -- a group is never partly active
    static const bool no_share = getenv("TG_BN_CONV1_NO_SHARE") != nullptr;
    const bool share = !no_share && (g.qpp == 16 || g.qpp == 32 || g.qpp == 64);
    if (share) hipLaunchKernelGGL(bn_bwd_conv1_reduce_kernel<true>, dim3(g.grid), dim3(256), 0, S(stream), f, cv, rows, C, g.qpp, g.rlanes, g.rows_per_block, ws);
    else hipLaunchKernelGGL(bn_bwd_conv1_reduce_kernel<false>, dim3(g.grid), dim3(256), 0, S(stream), f, cv, rows, C, g.qpp, g.rlanes, g.rows_per_block, ws);
    TG_CHECK_LAUNCH("bn_bwd_conv1_reduce_kernel");
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(fr_grid(C)), dim3(256), 0, S(stream), ws, g.grid, C, (double)rows, gamma, rstd, dgamma,
                       dbeta, dbias);
    TG_CHECK_LAUNCH("bn_bwd_final");
    // apply: contiguous row ranges per block as well (the coordinates advance incrementally), ~8 blocks per CU
    int64_t want = cdiv64(rows, (int64_t)g.rlanes * 32);
    if (want > 2048) want = 2048;
    if (want < 1) want = 1;
    const int64_t rpb = cdiv64(rows, want);
    const int agrid = (int)cdiv64(rows, rpb);
    if (share) hipLaunchKernelGGL(bn_bwd_conv1_apply_kernel<true>, dim3(agrid), dim3(256), 0, S(stream), f, cv, rows, C, g.qpp, g.rlanes, rpb, dgamma, dbeta, dy);
    else hipLaunchKernelGGL(bn_bwd_conv1_apply_kernel<false>, dim3(agrid), dim3(256), 0, S(stream), f, cv, rows, C, g.qpp, g.rlanes, rpb, dgamma, dbeta, dy);
    TG_CHECK_LAUNCH("bn_bwd_conv1_apply_kernel");
    return TG_OK;
}

extern "C" int tg_bn_act_bwd(const float* dout, const float* y, int64_t rows, int C, const float* mean, const float* rstd,
                             const float* gamma, const float* beta, int act, float slope, const float* ratio, float* dy,
                             float* dgamma, float* dbeta, float* dbias, float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(dout && y && mean && rstd && gamma && beta && dy && dgamma && dbeta && ws, "tg_bn_act_bwd: null pointer");
    TG_REQUIRE(C >= 1 && C <= 1024 && rows > 0, "tg_bn_act_bwd: bad dims");
    if (bn_small_ok(rows, C)) {
        hipLaunchKernelGGL(bn_bwd_small_kernel, dim3(C / 16), dim3(256), 5 * 4 * 64 * 4 * sizeof(double), S(stream), dout, y, (int)rows, C,
                           mean, rstd, gamma, beta, act, slope, ratio, dy, dgamma, dbeta, dbias);
        TG_CHECK_LAUNCH("bn_bwd_small_kernel");
        return TG_OK;
    }
    TG_REQUIRE(ws_bytes >= tg_bn_ws_bytes(rows, C), "tg_bn_act_bwd: workspace too small");
    BnBwdF f{dout, y, mean, rstd, gamma, beta, C, act, slope, ratio};
    int nblocks;
    if (C % 4 == 0) {
        ColGeom4 g = col_geom4(rows, C);
        nblocks = g.grid;
        hipLaunchKernelGGL((colreduce4_kernel<5, BnBwdF>), dim3(g.grid), dim3(256), 0, S(stream), f, rows, C, g.qpp, g.rlanes,
                           g.rows_per_block, ws, 0);
    } else {
        ColGeom g = col_geom(rows, C);
        nblocks = g.grid;
        hipLaunchKernelGGL((colreduce_kernel<5, BnBwdF>), dim3(g.grid), dim3(256), 0, S(stream), f, rows, C, g.cpp, g.rlanes,
                           g.rows_per_block, ws);
    }
    TG_CHECK_LAUNCH("bn_bwd_reduce");
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(fr_grid(C)), dim3(256), 0, S(stream), ws, nblocks, C, (double)rows, gamma, rstd,
                       dgamma, dbeta, dbias);
    TG_CHECK_LAUNCH("bn_bwd_final");
    if (C % 4 == 0) {
        RowGeom rg = row_geom(rows, C);
        hipLaunchKernelGGL(bn_bwd_apply4_kernel, rg.grid, dim3(256), 0, S(stream), dout, y, rows, C, rg.qpp, rg.rlanes, mean, rstd,
                           gamma, beta, act, slope, ratio, dgamma, dbeta, dy);
    } else {
        hipLaunchKernelGGL(bn_bwd_apply1_kernel, dim3(ew_grid(rows * C, 256)), dim3(256), 0, S(stream), dout, y, rows, C, mean,
                           rstd, gamma, beta, act, slope, ratio, dgamma, dbeta, dy);
    }
    TG_CHECK_LAUNCH("bn_bwd_apply");
    return TG_OK;
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                      int64_t rows, int C, int act, float slope,
                                                      const float* __restrict__ ratio, float* __restrict__ din) {
    const int64_t total = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float v = dout[i] * act_grad(out[i], act, slope);
        if (ratio) v *= ratio[i / C];
        din[i] = v;
    }
}
extern "C" int tg_act_bwd(const float* dout, const float* out, int64_t rows, int C, int act, float slope, const float* ratio,
                          float* din, tg_stream_t stream) {
    TG_REQUIRE(dout && din && rows > 0 && C > 0, "tg_act_bwd: bad arguments");
    TG_REQUIRE(out || act == TG_ACT_NONE, "tg_act_bwd: activation needs the forward output");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(rows * C, 256)), dim3(256), 0, S(stream), dout, out ? out : dout, rows, C,
                       act, slope, ratio, din);
    TG_CHECK_LAUNCH("act_bwd_kernel");
    return TG_OK;
}

// =================================================================================================
// bilinear x2 (align_corners=False) (+) concat, and its adjoint
// =================================================================================================
struct Lerp {
    int i0, i1;
    float l0, l1;
};
// PyTorch area_pixel_compute_source_index for scale 0.5, align_corners=False
__device__ __forceinline__ Lerp lerp_src(int dst, int n) {
    float src = 0.5f * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Lerp L;
    L.i0 = (int)src;
    L.i1 = L.i0 + (L.i0 < n - 1 ? 1 : 0);
    L.l1 = src - (float)L.i0;
    L.l0 = 1.f - L.l1;
    return L;
}

// the bilinear blend, every rounding spelled out (the generic and the x2 kernels must agree bit for bit, whatever the compiler
// would contract):  ly0 * (lx0 * a + lx1 * b) + ly1 * (lx0 * c + lx1 * d)
__device__ __forceinline__ float lerp_blend(float ly0, float ly1, float lx0, float lx1, float a, float b, float c, float d) {
    const float top = __fmaf_rn(lx1, b, __fmul_rn(lx0, a));
    const float bot = __fmaf_rn(lx1, d, __fmul_rn(lx0, c));
    return __fmaf_rn(ly1, bot, __fmul_rn(ly0, top));
}
// IDX = the type of the flat element index: uint32_t when the tensor has fewer than 2^31 vector elements (three 64-bit
// divisions per element made this gather ALU-bound: 3.4 TB/s)
template <int V, class IDX>  // V = 4 (float4 over channels) or 1
__global__ __launch_bounds__(256) void upcat_fwd_kernel(const float* __restrict__ up, const float* __restrict__ skip,
                                                        const float* __restrict__ omask, int B, int h, int w, int Cu, int H,
                                                        int W, int Cs, int offy, int offx, float* __restrict__ out) {
    const int Ct = Cu + Cs;
    const int cv = Ct / V;
    const IDX total = (IDX)B * H * W * cv;
    for (IDX idx = (IDX)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (IDX)gridDim.x * 256) {
        const int c = (int)(idx % (IDX)cv) * V;
        const IDX pixi = idx / (IDX)cv;
        const int X = (int)(pixi % (IDX)W);
        const IDX t = pixi / (IDX)W;
        const int Y = (int)(t % (IDX)H);
        const int b = (int)(t / (IDX)H);
        const int64_t pix = (int64_t)pixi;
        float o[V];
        if (c >= Cu) {
            if constexpr (V == 4) {
                const f32x4 t4 = *reinterpret_cast<const f32x4*>(skip + pix * Cs + (c - Cu));
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = t4[e];
            } else {
                o[0] = skip[pix * Cs + (c - Cu)];
            }
        } else {
            const int yu = Y - offy, xu = X - offx;
            if (yu < 0 || yu >= 2 * h || xu < 0 || xu >= 2 * w) {
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = 0.f;
            } else {
                const Lerp ly = lerp_src(yu, h), lx = lerp_src(xu, w);
                const float* r0 = up + (((int64_t)b * h + ly.i0) * w) * Cu + c;
                const float* r1 = up + (((int64_t)b * h + ly.i1) * w) * Cu + c;
                float va[V], vb[V], vc[V], vd[V];
                if constexpr (V == 4) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(r0 + (int64_t)lx.i0 * Cu), b4 = *reinterpret_cast<const f32x4*>(r0 + (int64_t)lx.i1 * Cu);
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(r1 + (int64_t)lx.i0 * Cu), d4 = *reinterpret_cast<const f32x4*>(r1 + (int64_t)lx.i1 * Cu);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { va[e] = a4[e]; vb[e] = b4[e]; vc[e] = c4[e]; vd[e] = d4[e]; }
                } else {
                    va[0] = r0[(int64_t)lx.i0 * Cu]; vb[0] = r0[(int64_t)lx.i1 * Cu];
                    vc[0] = r1[(int64_t)lx.i0 * Cu]; vd[0] = r1[(int64_t)lx.i1 * Cu];
                }
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = lerp_blend(ly.l0, ly.l1, lx.l0, lx.l1, va[e], vb[e], vc[e], vd[e]);
            }
        }
        const float mk = omask ? omask[pix] : 1.f;
        if constexpr (V == 4) {
            const f32x4 o4 = {__fmul_rn(o[0], mk), __fmul_rn(o[1], mk), __fmul_rn(o[2], mk), __fmul_rn(o[3], mk)};
            *reinterpret_cast<f32x4*>(out + pix * Ct + c) = o4;
        } else {
            out[pix * Ct + c] = __fmul_rn(o[0], mk);
        }
    }
}
// The exact x2 case (H = 2h, W = 2w, channel counts multiples of 4: every decoder level at the power-of-two tile sizes):
// one thread per SOURCE pixel and channel quad produces its 2 x 2 output pixels from the 3 x 3 source neighbourhood (9 16-byte
// loads for 4 outputs instead of 4 per output, two interpolation set-ups per axis instead of one per output element); the
// skip half of the concat is a second index range of the same launch.  Same interpolation expression, same operand order as
// upcat_fwd_kernel: bit-identical results (tests/test_hip_ops.py::test_upcat_x2_equals_generic).
// BNIN (tg_upcat_fwd_bn): `up` is the PRE-BatchNorm output of the decoder layer below and the interpolation runs over
// act(BN(up)), formed as the nine source quads arrive (bn_affine: the rounding sequence of bn_act_fwd, so the concat tensor equals
// the two-pass form bit for bit) -- that layer's activation is never written.
struct UpBn {
    const float *mean, *rstd, *gamma, *beta;
    int act;
    float slope;
};
template <bool BNIN>
__global__ __launch_bounds__(256) void upcat_fwd_x2_kernel(const float* __restrict__ up, const float* __restrict__ skip,
                                                           const float* __restrict__ omask, int B, int h, int w, int Cu, int Cs,
                                                           float* __restrict__ out, const UpBn bn) {
    const int Ct = Cu + Cs, H = 2 * h, W = 2 * w;
    const int cu4 = Cu >> 2, cs4 = Cs >> 2;
    const uint32_t nA = (uint32_t)B * h * w * cu4, nB = (uint32_t)B * H * W * cs4;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < nA + nB; idx += gridDim.x * 256u) {
        if (idx < nA) {
            const int c = (int)(idx % (uint32_t)cu4) * 4;
            uint32_t t = idx / (uint32_t)cu4;
            const int x = (int)(t % (uint32_t)w);
            t /= (uint32_t)w;
            const int y = (int)(t % (uint32_t)h), b = (int)(t / (uint32_t)h);
            const int ry[3] = {y > 0 ? y - 1 : 0, y, y < h - 1 ? y + 1 : h - 1};
            const int rx[3] = {x > 0 ? x - 1 : 0, x, x < w - 1 ? x + 1 : w - 1};
            f32x4 T[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    T[i][j] = *reinterpret_cast<const f32x4*>(up + (((size_t)b * h + ry[i]) * w + rx[j]) * Cu + c);
            if constexpr (BNIN) {
                const f32x4 mv = *reinterpret_cast<const f32x4*>(bn.mean + c), rv = *reinterpret_cast<const f32x4*>(bn.rstd + c);
                const f32x4 gv = *reinterpret_cast<const f32x4*>(bn.gamma + c), bv = *reinterpret_cast<const f32x4*>(bn.beta + c);
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) T[i][j][e] = apply_act(bn_affine(T[i][j][e], mv[e], rv[e], gv[e], bv[e]), bn.act, bn.slope);
            }
            Lerp ly[2], lx[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                ly[a] = lerp_src(2 * y + a, h);
                lx[a] = lerp_src(2 * x + a, w);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int i0 = ly[a].i0 - y + 1, i1 = ly[a].i1 - y + 1;          // in {0, 1, 2}: rows y-1, y, y+1 (clamped like ry)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int j0 = lx[bb].i0 - x + 1, j1 = lx[bb].i1 - x + 1;
                    // register arrays indexed by computed values: select instead (i0 in {0,1}, i1 in {1,2})
                    const f32x4 a00 = i0 == 0 ? (j0 == 0 ? T[0][0] : T[0][1]) : (j0 == 0 ? T[1][0] : T[1][1]);
                    const f32x4 a01 = i0 == 0 ? (j1 == 1 ? T[0][1] : T[0][2]) : (j1 == 1 ? T[1][1] : T[1][2]);
                    const f32x4 a10 = i1 == 1 ? (j0 == 0 ? T[1][0] : T[1][1]) : (j0 == 0 ? T[2][0] : T[2][1]);
                    const f32x4 a11 = i1 == 1 ? (j1 == 1 ? T[1][1] : T[1][2]) : (j1 == 1 ? T[2][1] : T[2][2]);
                    const size_t pix = ((size_t)b * H + 2 * y + a) * W + 2 * x + bb;
                    const float mk = omask ? omask[pix] : 1.f;
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        o[e] = __fmul_rn(lerp_blend(ly[a].l0, ly[a].l1, lx[bb].l0, lx[bb].l1, a00[e], a01[e], a10[e], a11[e]), mk);
                    *reinterpret_cast<f32x4*>(out + pix * Ct + c) = o;
                }
            }
        } else {
            const uint32_t k = idx - nA;
            const int c = (int)(k % (uint32_t)cs4) * 4;
            const size_t pix = k / (uint32_t)cs4;
            const float mk = omask ? omask[pix] : 1.f;
            const f32x4 v = *reinterpret_cast<const f32x4*>(skip + pix * Cs + c);
            *reinterpret_cast<f32x4*>(out + pix * Ct + Cu + c) = f32x4{__fmul_rn(v[0], mk), __fmul_rn(v[1], mk), __fmul_rn(v[2], mk), __fmul_rn(v[3], mk)};
        }
    }
}
static bool upcat_x2_ok(int B, int h, int w, int Cu, int H, int W, int Cs) {
    const bool no_x2 = getenv("TG_NO_UPCAT_X2") != nullptr;      // read per call: tests flip it at run time
    return !no_x2 && (int64_t)B * H * W * (Cu + Cs) < ((int64_t)1 << 31) && H == 2 * h && W == 2 * w && Cu % 4 == 0 && Cs % 4 == 0;
}
extern "C" int tg_upcat_bn_supported(int B, int h, int w, int Cu, int H, int W, int Cs) {
    return B > 0 && h > 0 && w > 0 && Cu > 0 && Cs >= 0 && upcat_x2_ok(B, h, w, Cu, H, W, Cs) ? 1 : 0;
}
extern "C" int tg_upcat_fwd_bn(const float* up, const TgBnAct* bn, const float* skip, const float* out_mask, int B, int h, int w, int Cu,
                               int H, int W, int Cs, float* out, tg_stream_t stream) {
    TG_REQUIRE(up && out && bn && bn->mean && bn->rstd && bn->gamma && bn->beta, "tg_upcat_fwd_bn: null pointer");
    TG_REQUIRE(Cs == 0 || skip, "tg_upcat_fwd_bn: skip is NULL but Cs > 0");
    TG_REQUIRE(tg_upcat_bn_supported(B, h, w, Cu, H, W, Cs), "tg_upcat_fwd_bn: geometry not supported (ask tg_upcat_bn_supported first)");
    const int64_t items = (int64_t)B * h * w * (Cu / 4) + (int64_t)B * H * W * (Cs / 4);
    const UpBn ub{bn->mean, bn->rstd, bn->gamma, bn->beta, bn->act, bn->slope};
    hipLaunchKernelGGL(upcat_fwd_x2_kernel<true>, dim3(ew_grid(items, 256)), dim3(256), 0, S(stream), up, skip, out_mask, B, h, w, Cu, Cs, out, ub);
    TG_CHECK_LAUNCH("upcat_fwd_x2_kernel");
    return TG_OK;
}
extern "C" int tg_upcat_fwd(const float* up, const float* skip, const float* out_mask, int B, int h, int w, int Cu, int H, int W,
                            int Cs, float* out, tg_stream_t stream) {
    TG_REQUIRE(up && out && B > 0 && h > 0 && w > 0 && Cu > 0 && H > 0 && W > 0 && Cs >= 0, "tg_upcat_fwd: bad arguments");
    TG_REQUIRE(Cs == 0 || skip, "tg_upcat_fwd: skip is NULL but Cs > 0");
    const int offy = floordiv2(H - 2 * h), offx = floordiv2(W - 2 * w);
    const int Ct = Cu + Cs;
    const bool small = (int64_t)B * H * W * Ct < ((int64_t)1 << 31);
    if (upcat_x2_ok(B, h, w, Cu, H, W, Cs)) {
        const int64_t items = (int64_t)B * h * w * (Cu / 4) + (int64_t)B * H * W * (Cs / 4);
        hipLaunchKernelGGL(upcat_fwd_x2_kernel<false>, dim3(ew_grid(items, 256)), dim3(256), 0, S(stream), up, skip, out_mask, B, h, w, Cu, Cs, out,
                           UpBn{});
        TG_CHECK_LAUNCH("upcat_fwd_x2_kernel");
        return TG_OK;
    }
    if (Cu % 4 == 0 && Cs % 4 == 0) {
        auto kern = small ? upcat_fwd_kernel<4, uint32_t> : upcat_fwd_kernel<4, int64_t>;
        hipLaunchKernelGGL(kern, dim3(ew_grid((int64_t)B * H * W * (Ct / 4), 256)), dim3(256), 0, S(stream), up, skip, out_mask, B, h, w,
                           Cu, H, W, Cs, offy, offx, out);
    } else {
        auto kern = small ? upcat_fwd_kernel<1, uint32_t> : upcat_fwd_kernel<1, int64_t>;
        hipLaunchKernelGGL(kern, dim3(ew_grid((int64_t)B * H * W * Ct, 256)), dim3(256), 0, S(stream), up, skip, out_mask, B, h, w, Cu, H,
                           W, Cs, offy, offx, out);
    }
    TG_CHECK_LAUNCH("upcat_fwd_kernel");
    return TG_OK;
}

template <int V>
__global__ __launch_bounds__(256) void upcat_bwd_up_kernel(const float* __restrict__ dout, int B, int h, int w, int Cu, int H,
                                                           int W, int Ct, int offy, int offx, float* __restrict__ dup) {
    const int cv = Cu / V;
    const int64_t total = (int64_t)B * h * w * cv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % cv) * V;
        const int64_t pix = idx / cv;
        const int x = (int)(pix % w);
        const int64_t t = pix / w;
        const int y = (int)(t % h);
        const int b = (int)(t / h);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        for (int yu = 2 * y - 1; yu <= 2 * y + 2; ++yu) {
            const int Y = yu + offy;
            if (yu < 0 || yu >= 2 * h || Y < 0 || Y >= H) continue;
            const Lerp ly = lerp_src(yu, h);
            const float wy = (ly.i0 == y ? ly.l0 : 0.f) + (ly.i1 == y ? ly.l1 : 0.f);
            if (wy == 0.f) continue;
            for (int xu = 2 * x - 1; xu <= 2 * x + 2; ++xu) {
                const int X = xu + offx;
                if (xu < 0 || xu >= 2 * w || X < 0 || X >= W) continue;
                const Lerp lx = lerp_src(xu, w);
                const float wx = (lx.i0 == x ? lx.l0 : 0.f) + (lx.i1 == x ? lx.l1 : 0.f);
                if (wx == 0.f) continue;
                const float* src = dout + (((int64_t)b * H + Y) * W + X) * Ct + c;
                if constexpr (V == 4) {
                    const f32x4 s4 = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = __fmaf_rn(__fmul_rn(wy, wx), s4[e], acc[e]);
                } else {
                    acc[0] = __fmaf_rn(__fmul_rn(wy, wx), src[0], acc[0]);
                }
            }
        }
        if constexpr (V == 4) {
            const f32x4 a4 = {acc[0], acc[1], acc[2], acc[3]};
            *reinterpret_cast<f32x4*>(dup + pix * Cu + c) = a4;
        } else {
            dup[pix * Cu + c] = acc[0];
        }
    }
}
__global__ __launch_bounds__(256) void slice_channels_kernel(const float* __restrict__ src, int64_t rows, int Ct, int c0,
                                                             int Cs, float* __restrict__ dst) {
    const int64_t total = rows * Cs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cs);
        const int64_t r = i / Cs;
        dst[i] = src[r * Ct + c0 + c];
    }
}
// x2 case of the adjoint (see upcat_fwd_x2_kernel): the four row / column weights of the 4 x 4 output neighbourhood are set up
// once per thread (8 interpolation set-ups instead of 20), accumulated in the generic kernel's order (bit-identical); the skip
// half of the gradient (a channel slice of dout) is a second index range of the same launch instead of a second kernel.
__global__ __launch_bounds__(256) void upcat_bwd_x2_kernel(const float* __restrict__ dout, int B, int h, int w, int Cu, int Cs,
                                                           float* __restrict__ dup, float* __restrict__ dskip) {
    const int Ct = Cu + Cs, H = 2 * h, W = 2 * w;
    const int cu4 = Cu >> 2, cs4 = dskip ? Cs >> 2 : 0;
    const uint32_t nA = (uint32_t)B * h * w * cu4, nB = (uint32_t)B * H * W * cs4;
    for (uint32_t idx = blockIdx.x * 256u + threadIdx.x; idx < nA + nB; idx += gridDim.x * 256u) {
        if (idx < nA) {
            const int c = (int)(idx % (uint32_t)cu4) * 4;
            uint32_t t = idx / (uint32_t)cu4;
            const int x = (int)(t % (uint32_t)w);
            t /= (uint32_t)w;
            const int y = (int)(t % (uint32_t)h), b = (int)(t / (uint32_t)h);
            float wy[4], wx[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yu = 2 * y - 1 + k, xu = 2 * x - 1 + k;
                wy[k] = 0.f;
                wx[k] = 0.f;
                if (yu >= 0 && yu < H) {
                    const Lerp l = lerp_src(yu, h);
                    wy[k] = (l.i0 == y ? l.l0 : 0.f) + (l.i1 == y ? l.l1 : 0.f);
                }
                if (xu >= 0 && xu < W) {
                    const Lerp l = lerp_src(xu, w);
                    wx[k] = (l.i0 == x ? l.l0 : 0.f) + (l.i1 == x ? l.l1 : 0.f);
                }
            }
            // all sixteen quads requested together at clamped coordinates (a weight is zero only on the border, where the clamped
            // quad is dropped by the select below): with the load inside `if (w != 0)` every quad was awaited by its own FMA before
            // the next branch could issue -- sixteen round trips in a row per thread
            f32x4 S[4][4];
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int yu = min(max(2 * y - 1 + ky, 0), H - 1);
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const int xu = min(max(2 * x - 1 + kx, 0), W - 1);
                    S[ky][kx] = *reinterpret_cast<const f32x4*>(dout + (((size_t)b * H + yu) * W + xu) * Ct + c);
                }
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 4; ++ky)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    const bool on = wy[ky] != 0.f && wx[kx] != 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = on ? __fmaf_rn(__fmul_rn(wy[ky], wx[kx]), S[ky][kx][e], acc[e]) : acc[e];
                }
            *reinterpret_cast<f32x4*>(dup + (((size_t)b * h + y) * w + x) * Cu + c) = acc;
        } else {
            const uint32_t k = idx - nA;
            const int c = (int)(k % (uint32_t)cs4) * 4;
            const size_t pix = k / (uint32_t)cs4;
            *reinterpret_cast<f32x4*>(dskip + pix * Cs + c) = *reinterpret_cast<const f32x4*>(dout + pix * Ct + Cu + c);
        }
    }
}
extern "C" int tg_upcat_bwd(const float* dout, int B, int h, int w, int Cu, int H, int W, int Cs, float* dup, float* dskip,
                            tg_stream_t stream) {
    TG_REQUIRE(dout && dup && B > 0 && h > 0 && w > 0 && Cu > 0 && H > 0 && W > 0 && Cs >= 0, "tg_upcat_bwd: bad arguments");
    const int offy = floordiv2(H - 2 * h), offx = floordiv2(W - 2 * w);
    const int Ct = Cu + Cs;
    const bool no_x2 = getenv("TG_NO_UPCAT_X2") != nullptr;      // read per call: tests flip it at run time
    if (!no_x2 && H == 2 * h && W == 2 * w && Cu % 4 == 0 && Cs % 4 == 0 && (int64_t)B * H * W * Ct < ((int64_t)1 << 31)) {
        const int64_t items = (int64_t)B * h * w * (Cu / 4) + (dskip ? (int64_t)B * H * W * (Cs / 4) : 0);
        hipLaunchKernelGGL(upcat_bwd_x2_kernel, dim3(ew_grid(items, 256)), dim3(256), 0, S(stream), dout, B, h, w, Cu, Cs, dup,
                           Cs > 0 ? dskip : nullptr);
        TG_CHECK_LAUNCH("upcat_bwd_x2_kernel");
        return TG_OK;
    }
    if (Cu % 4 == 0 && Ct % 4 == 0) {
        hipLaunchKernelGGL((upcat_bwd_up_kernel<4>), dim3(ew_grid((int64_t)B * h * w * (Cu / 4), 256)), dim3(256), 0, S(stream),
                           dout, B, h, w, Cu, H, W, Ct, offy, offx, dup);
    } else {
        hipLaunchKernelGGL((upcat_bwd_up_kernel<1>), dim3(ew_grid((int64_t)B * h * w * Cu, 256)), dim3(256), 0, S(stream), dout,
                           B, h, w, Cu, H, W, Ct, offy, offx, dup);
    }
    TG_CHECK_LAUNCH("upcat_bwd_up_kernel");
    if (Cs > 0 && dskip) {
        hipLaunchKernelGGL(slice_channels_kernel, dim3(ew_grid((int64_t)B * H * W * Cs, 256)), dim3(256), 0, S(stream), dout,
                           (int64_t)B * H * W, Ct, Cu, Cs, dskip);
        TG_CHECK_LAUNCH("slice_channels_kernel");
    }
    return TG_OK;
}

// =================================================================================================
// generator head
// =================================================================================================
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void sigcomp_fwd_kernel(const float* __restrict__ z, const float* __restrict__ x,
                                                          const float* __restrict__ m, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = sigmoidf_(z[i]);
        out[i] = s * (1.f - m[i]) + x[i] * m[i];
    }
}
extern "C" int tg_sigmoid_composite_fwd(const float* logits, const float* x, const float* mask, int64_t n, float* out,
                                        tg_stream_t stream) {
    TG_REQUIRE(logits && x && mask && out && n > 0, "tg_sigmoid_composite_fwd: bad arguments");
    hipLaunchKernelGGL(sigcomp_fwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), logits, x, mask, n, out);
    TG_CHECK_LAUNCH("sigcomp_fwd_kernel");
    return TG_OK;
}
__global__ __launch_bounds__(256) void sigcomp_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ z,
                                                          const float* __restrict__ m, int64_t n, float* __restrict__ dz,
                                                          float* __restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = sigmoidf_(z[i]);
        float g = dout[i];
        dz[i] = g * (1.f - m[i]) * (1.f - s) * s;
        if (dx) dx[i] = g * m[i];
    }
}
extern "C" int tg_sigmoid_composite_bwd(const float* dout, const float* logits, const float* mask, int64_t n, float* dlogits,
                                        float* dx, tg_stream_t stream) {
    TG_REQUIRE(dout && logits && mask && dlogits && n > 0, "tg_sigmoid_composite_bwd: bad arguments");
    hipLaunchKernelGGL(sigcomp_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), dout, logits, mask, n, dlogits, dx);
    TG_CHECK_LAUNCH("sigcomp_bwd_kernel");
    return TG_OK;
}

// =================================================================================================
// 2x2 max-pool (VGG trunk)
// =================================================================================================
template <int V>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int C,
                                                           float* __restrict__ out) {
    const int Ho = H / 2, Wo = W / 2;
    const int cv = C / V;
    const int64_t total = (int64_t)B * Ho * Wo * cv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % cv) * V;
        int64_t t = idx / cv;
        const int ox = (int)(t % Wo);
        t /= Wo;
        const int oy = (int)(t % Ho);
        const int b = (int)(t / Ho);
        const float* p0 = x + (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float v = p0[e];
            v = fmaxf(v, p0[C + e]);
            v = fmaxf(v, p0[(int64_t)W * C + e]);
            v = fmaxf(v, p0[(int64_t)W * C + C + e]);
            out[(((int64_t)b * Ho + oy) * Wo + ox) * C + c + e] = v;
        }
    }
}
extern "C" int tg_maxpool2_fwd(const float* x, int B, int H, int W, int C, float* out, tg_stream_t stream) {
    TG_REQUIRE(x && out && B > 0 && H > 1 && W > 1 && C > 0, "tg_maxpool2_fwd: bad arguments");
    if (C % 4 == 0)
        hipLaunchKernelGGL((maxpool2_fwd_kernel<4>), dim3(ew_grid((int64_t)B * (H / 2) * (W / 2) * C / 4, 256)), dim3(256), 0,
                           S(stream), x, B, H, W, C, out);
    else
        hipLaunchKernelGGL((maxpool2_fwd_kernel<1>), dim3(ew_grid((int64_t)B * (H / 2) * (W / 2) * C, 256)), dim3(256), 0, S(stream),
                           x, B, H, W, C, out);
    TG_CHECK_LAUNCH("maxpool2_fwd_kernel");
    return TG_OK;
}
// gradient goes to the first maximum in window scan order (ATen CPU max_pool2d); one thread per pooling window and
// V channels writes the window's four input gradients (odd trailing row/column, which no window covers, get zero)
template <int V>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x, int B,
                                                           int H, int W, int C, int relu_gate, float* __restrict__ dx) {
    const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, Ho = H / 2, Wo = W / 2;
    const int cv = C / V;
    const int64_t total = (int64_t)B * Hc * Wc * cv;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % cv) * V;
        int64_t t = idx / cv;
        const int ox = (int)(t % Wc);
        t /= Wc;
        const int oy = (int)(t % Hc);
        const int b = (int)(t / Hc);
        const int64_t base = (((int64_t)b * H + 2 * oy) * W + 2 * ox) * C + c;
        const bool full = oy < Ho && ox < Wo;
        const bool has_r = 2 * ox + 1 < W, has_d = 2 * oy + 1 < H;
        float xv[4][V], gv[4][V], dv[V];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < V; ++e) { xv[k][e] = 0.f; gv[k][e] = 0.f; }
        if (full) {
            const int64_t offs[4] = {0, C, (int64_t)W * C, (int64_t)W * C + C};
            if constexpr (V == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(x + base + offs[k]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[k][e] = t4[e];
                }
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dout + (((int64_t)b * Ho + oy) * Wo + ox) * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[e] = d4[e];
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) xv[k][0] = x[base + offs[k]];
                dv[0] = dout[(((int64_t)b * Ho + oy) * Wo + ox) * C + c];
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                int am = 0;
                float best = xv[0][e];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (xv[k][e] > best) { best = xv[k][e]; am = k; }
                const float gval = (relu_gate && best <= 0.f) ? 0.f : dv[e];
#pragma unroll
                for (int k = 0; k < 4; ++k) gv[k][e] = (k == am) ? gval : 0.f;
            }
        }
        const int64_t offs2[4] = {0, C, (int64_t)W * C, (int64_t)W * C + C};
        const bool wr[4] = {true, has_r, has_d, has_r && has_d};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!wr[k]) continue;
            if constexpr (V == 4) {
                f32x4 o = {gv[k][0], gv[k][1], gv[k][2], gv[k][3]};
                *reinterpret_cast<f32x4*>(dx + base + offs2[k]) = o;
            } else {
                dx[base + offs2[k]] = gv[k][0];
            }
        }
    }
}
// backward of the fused conv -> ReLU -> max-pool from the pool's CODE (tg_conv_fwd_pool_code: bits 0-1 window position of the
// maximum, bit 2 "maximum > 0"): one thread per pooled pixel and channel quad reads 16 B of gradient + 4 B of code and writes the
// window's four quads -- the full-resolution activation is not read (it was never written)
__global__ __launch_bounds__(256) void maxpool2_bwd_code_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ code, int B,
                                                                int Ho, int Wo, int C, float* __restrict__ dx) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)B * Ho * Wo * c4n;
    const int64_t rowpitch = (int64_t)2 * Wo * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % c4n) * 4;
        const int64_t pp = idx / c4n;                 // pooled pixel (b, oy, ox)
        const int ox = (int)(pp % Wo);
        const int64_t t = pp / Wo;
        const int oy = (int)(t % Ho), b = (int)(t / Ho);
        const f32x4 d = *reinterpret_cast<const f32x4*>(dout + pp * C + c);
        const uint32_t cd = *reinterpret_cast<const uint32_t*>(code + pp * C + c);
        float* base = dx + (((int64_t)b * 2 * Ho + 2 * oy) * 2 * Wo + 2 * ox) * C + c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t ce = (cd >> (8 * e)) & 0xffu;
                o[e] = ((ce & 3u) == (uint32_t)k && (ce & 4u)) ? d[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(base + (k >> 1) * rowpitch + (k & 1) * C) = o;
        }
    }
}
extern "C" int tg_maxpool2_bwd_code(const float* dout, const unsigned char* code, int B, int Ho, int Wo, int C, float* dx, tg_stream_t stream) {
    TG_REQUIRE(dout && code && dx && B > 0 && Ho > 0 && Wo > 0 && C > 0 && (C % 4) == 0, "tg_maxpool2_bwd_code: bad arguments");
    hipLaunchKernelGGL(maxpool2_bwd_code_kernel, dim3(ew_grid((int64_t)B * Ho * Wo * (C / 4), 256)), dim3(256), 0, S(stream), dout, code, B, Ho,
                       Wo, C, dx);
    TG_CHECK_LAUNCH("maxpool2_bwd_code_kernel");
    return TG_OK;
}
extern "C" int tg_maxpool2_bwd(const float* dout, const float* x, int B, int H, int W, int C, int relu_gate, float* dx,
                               tg_stream_t stream) {
    TG_REQUIRE(dout && x && dx && B > 0 && H > 1 && W > 1 && C > 0, "tg_maxpool2_bwd: bad arguments");
    const int64_t windows = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
    if (C % 4 == 0)
        hipLaunchKernelGGL((maxpool2_bwd_kernel<4>), dim3(ew_grid(windows * C / 4, 256)), dim3(256), 0, S(stream), dout, x, B, H, W, C,
                           relu_gate, dx);
    else
        hipLaunchKernelGGL((maxpool2_bwd_kernel<1>), dim3(ew_grid(windows * C, 256)), dim3(256), 0, S(stream), dout, x, B, H, W, C,
                           relu_gate, dx);
    TG_CHECK_LAUNCH("maxpool2_bwd_kernel");
    return TG_OK;
}

// =================================================================================================
// scalar reductions: per-block fp64 partials -> one finalize thread block
// =================================================================================================
template <int NQ>
__device__ __forceinline__ void block_reduce_store(double (&q)[NQ], double* __restrict__ partial) {
    __shared__ double red[NQ][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        double v = wave_sum_d(q[i]);
        if (lane == 0) red[i][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) partial[(size_t)blockIdx.x * NQ + i] = red[i][0] + red[i][1] + red[i][2] + red[i][3];
    }
}
static int red_grid(int64_t n) {
    int64_t g = cdiv64(n, 256 * 8);
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    return (int)g;
}
extern "C" size_t tg_reduce_ws_bytes(int64_t n) { return (size_t)(red_grid(n) * 8 + 16) * sizeof(double); }
extern "C" size_t tg_pixel_loss_ws_bytes(int B, int H, int W) { return tg_reduce_ws_bytes((int64_t)B * H * W); }

// ---- pixel-space losses -------------------------------------------------------------------------------
__device__ __forceinline__ float band_at(const float* __restrict__ m, int b, int y, int x, int H, int W) {
    // nine unconditional loads at CLAMPED coordinates, all in flight: a clamped tap repeats a pixel that is inside the window
    // anyway, so the maximum and the minimum over the in-range taps are unchanged (the rolled loop with its `continue`s awaited
    // one load after the other: 29 us for a 1 M-pixel pass)
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = min(max(y + t / 3 - 1, 0), H - 1), xx = min(max(x + t % 3 - 1, 0), W - 1);
        v[t] = m[((int64_t)b * H + yy) * W + xx];
    }
    float mx = v[0], mn = v[0];
#pragma unroll
    for (int t = 1; t < 9; ++t) {
        mx = fmaxf(mx, v[t]);
        mn = fminf(mn, v[t]);
    }
    // dilated - eroded, eroded = 1 - maxpool(1 - m) = min(m) over the window (losses.py:406-408)
    float d = mx - (1.f - (1.f - mn));
    return fminf(fmaxf(d, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void pixel_loss_sums_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                              const float* __restrict__ m, const float* __restrict__ l1w,
                                                              int B, int H, int W, double* __restrict__ partial) {
    const int64_t total = (int64_t)B * H * W;
    double q[5] = {0, 0, 0, 0, 0};
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int x = (int)(idx % W);
        const int64_t tt = idx / W;
        const int y = (int)(tt % H);
        const int b = (int)(tt / H);
        const float ad = fabsf(p[idx] - t[idx]);
        q[0] += (double)(l1w ? ad * l1w[idx] : ad);
        const float xh = p[idx] * (1.f - m[idx]);
        if (y > 0) {
            float d = xh - p[idx - W] * (1.f - m[idx - W]);
            q[1] += (double)(d * d);
        }
        if (x > 0) {
            float d = xh - p[idx - 1] * (1.f - m[idx - 1]);
            q[2] += (double)(d * d);
        }
        const float bd = band_at(m, b, y, x, H, W);
        q[3] += (double)(ad * bd);
        q[4] += (double)bd;
    }
    block_reduce_store<5>(q, partial);
}
// out5 = {l1, tv, boundary, sum(band), total}; coef[0..3] = per-element gradient coefficients
__global__ void pixel_loss_final_kernel(const double* __restrict__ partial, int nblocks, int B, int H, int W, float w_l1,
                                        float w_tv, float w_bnd, float eps, const float* __restrict__ gscale,
                                        float* __restrict__ out5, float* __restrict__ coef) {
    // one wave: lane l sums blocks l, l+64, ... in order, then a fixed xor tree over the lanes -> deterministic
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 64)
#pragma unroll
        for (int i = 0; i < 5; ++i) s[i] += partial[(size_t)b * 5 + i];
#pragma unroll
    for (int i = 0; i < 5; ++i) s[i] = wave_sum_d(s[i]);
    if (threadIdx.x != 0) return;
    const double n = (double)B * H * W;
    const double count_h = (double)B * (H - 1) * W, count_w = (double)B * H * (W - 1);
    const float l1 = (float)(s[0] / n);
    const float h_tv = (float)s[1], w_tv_sum = (float)s[2];
    // losses.py:127: 2 * (h_tv / count_h + w_tv / count_w) / batch_size
    const float tv = 2.f * (h_tv / (float)count_h + w_tv_sum / (float)count_w) / (float)B;
    const float den = (float)s[4];
    float bnd = 0.f;
    bool bnd_on = den >= 1.0f;
    if (bnd_on) {
        bnd = (float)s[3] / (den + eps);
        if (isnan(bnd) || isinf(bnd)) { bnd = 0.f; bnd_on = false; }
    }
    out5[0] = l1;
    out5[1] = tv;
    out5[2] = bnd;
    out5[3] = den;
    out5[4] = w_l1 * l1 + w_tv * tv + w_bnd * bnd;
    const float gs = gscale ? *gscale : 1.f;
    coef[0] = gs * w_l1 / (float)n;
    coef[1] = gs * w_tv * 2.f / ((float)B * (float)count_h) * 2.f;   // d/dx of d^2 carries the second 2
    coef[2] = gs * w_tv * 2.f / ((float)B * (float)count_w) * 2.f;
    coef[3] = bnd_on ? gs * w_bnd / (den + eps) : 0.f;
}
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) - (v < 0.f); }
__global__ __launch_bounds__(256) void pixel_loss_grad_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                              const float* __restrict__ m, const float* __restrict__ l1w,
                                                              int B, int H, int W, const float* __restrict__ coef,
                                                              int accumulate, float* __restrict__ dp) {
    const int64_t total = (int64_t)B * H * W;
    const float c_l1 = coef[0], c_h = coef[1], c_w = coef[2], c_b = coef[3];
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int x = (int)(idx % W);
        const int64_t tt = idx / W;
        const int y = (int)(tt % H);
        const int b = (int)(tt / H);
        const float s = sgn(p[idx] - t[idx]);
        float g = c_l1 * s * (l1w ? l1w[idx] : 1.f);
        const float hm = 1.f - m[idx];
        const float xh = p[idx] * hm;
        float dh = 0.f, dw = 0.f;
        if (y > 0) dh += xh - p[idx - W] * (1.f - m[idx - W]);
        if (y < H - 1) dh -= p[idx + W] * (1.f - m[idx + W]) - xh;
        if (x > 0) dw += xh - p[idx - 1] * (1.f - m[idx - 1]);
        if (x < W - 1) dw -= p[idx + 1] * (1.f - m[idx + 1]) - xh;
        g += hm * (c_h * dh + c_w * dw);
        if (c_b != 0.f) g += c_b * s * band_at(m, b, y, x, H, W);
        dp[idx] = accumulate ? dp[idx] + g : g;
    }
}
extern "C" int tg_pixel_losses(const float* pred, const float* target, const float* mask, const float* l1_weight, int B,
                               int H, int W, float w_l1, float w_tv, float w_bnd, float bnd_eps, const float* gscale,
                               float* out5, float* dpred, int accumulate, float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(pred && target && mask && out5 && ws, "tg_pixel_losses: null pointer");
    TG_REQUIRE(B > 0 && H > 1 && W > 1, "tg_pixel_losses: bad dims");
    TG_REQUIRE(ws_bytes >= tg_pixel_loss_ws_bytes(B, H, W), "tg_pixel_losses: workspace too small");
    const int64_t n = (int64_t)B * H * W;
    const int grid = red_grid(n);
    double* partial = reinterpret_cast<double*>(ws);
    float* coef = reinterpret_cast<float*>(partial + (size_t)grid * 8);
    hipLaunchKernelGGL(pixel_loss_sums_kernel, dim3(grid), dim3(256), 0, S(stream), pred, target, mask, l1_weight, B, H, W,
                       partial);
    TG_CHECK_LAUNCH("pixel_loss_sums_kernel");
    hipLaunchKernelGGL(pixel_loss_final_kernel, dim3(1), dim3(64), 0, S(stream), partial, grid, B, H, W, w_l1, w_tv, w_bnd,
                       bnd_eps, gscale, out5, coef);
    TG_CHECK_LAUNCH("pixel_loss_final_kernel");
    if (dpred) {
        hipLaunchKernelGGL(pixel_loss_grad_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), pred, target, mask, l1_weight,
                           B, H, W, coef, accumulate, dpred);
        TG_CHECK_LAUNCH("pixel_loss_grad_kernel");
    }
    return TG_OK;
}
// ---- mean |a-b| (+ gradient) -----------------------------------------------------------------------------
// RELU_GATE: `a` is a ReLU output and the gradient wanted is the one in front of that ReLU: da = (a > 0) * d mean|a-b| / da
template <bool RELU_GATE>
__global__ __launch_bounds__(256) void l1_mean_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                      float coef, const float* __restrict__ gscale, float* __restrict__ da,
                                                      double* __restrict__ partial) {
    double q[1] = {0};
    const float k = da ? coef * (gscale ? *gscale : 1.f) / (float)n : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float av = a[i];
        const float d = av - b[i];
        q[0] += (double)fabsf(d);
        if (da) da[i] = (!RELU_GATE || av > 0.f) ? k * sgn(d) : 0.f;
    }
    block_reduce_store<1>(q, partial);
}
__global__ void mean_final_kernel(const double* __restrict__ partial, int nblocks, double inv_n, float* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    double s = 0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[b];
    s = wave_sum_d(s);
    if (threadIdx.x == 0) out[0] = (float)(s * inv_n);
}
static int l1_mean_launch(const float* a, const float* b, int64_t n, float coef, const float* gscale, float* out1, float* da,
                          float* ws, size_t ws_bytes, tg_stream_t stream, bool relu_gate) {
    TG_REQUIRE(a && b && out1 && ws && n > 0, "tg_l1_mean: bad arguments");
    TG_REQUIRE(ws_bytes >= tg_reduce_ws_bytes(n), "tg_l1_mean: workspace too small");
    const int grid = red_grid(n);
    double* partial = reinterpret_cast<double*>(ws);
    if (relu_gate) hipLaunchKernelGGL(l1_mean_kernel<true>, dim3(grid), dim3(256), 0, S(stream), a, b, n, coef, gscale, da, partial);
    else hipLaunchKernelGGL(l1_mean_kernel<false>, dim3(grid), dim3(256), 0, S(stream), a, b, n, coef, gscale, da, partial);
    TG_CHECK_LAUNCH("l1_mean_kernel");
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(64), 0, S(stream), partial, grid, 1.0 / (double)n, out1);
    TG_CHECK_LAUNCH("mean_final_kernel");
    return TG_OK;
}
extern "C" int tg_l1_mean(const float* a, const float* b, int64_t n, float coef, const float* gscale, float* out1, float* da,
                          float* ws, size_t ws_bytes, tg_stream_t stream) {
    return l1_mean_launch(a, b, n, coef, gscale, out1, da, ws, ws_bytes, stream, false);
}
extern "C" int tg_l1_mean_relu(const float* a, const float* b, int64_t n, float coef, const float* gscale, float* out1, float* da,
                               float* ws, size_t ws_bytes, tg_stream_t stream) {
    return l1_mean_launch(a, b, n, coef, gscale, out1, da, ws, ws_bytes, stream, true);
}

// ---- BCE with logits, constant target -------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ z, int64_t n, float target, float coef,
                                                  const float* __restrict__ gscale, float* __restrict__ dz,
                                                  double* __restrict__ partial) {
    double q[1] = {0};
    const float k = dz ? coef * (gscale ? *gscale : 1.f) / (float)n : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = z[i];
        // ATen: (1 - t) * x + max(-x, 0) + log(exp(-max) + exp(-x - max))  ==  max(x,0) - x*t + log1p(exp(-|x|))
        const float loss = fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x)));
        q[0] += (double)loss;
        if (dz) dz[i] = k * (sigmoidf_(x) - target);
    }
    block_reduce_store<1>(q, partial);
}
extern "C" int tg_bce_logits(const float* z, int64_t n, float target, float coef, const float* gscale, float* out1, float* dz,
                             float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(z && out1 && ws && n > 0, "tg_bce_logits: bad arguments");
    TG_REQUIRE(ws_bytes >= tg_reduce_ws_bytes(n), "tg_bce_logits: workspace too small");
    const int grid = red_grid(n);
    double* partial = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(bce_kernel, dim3(grid), dim3(256), 0, S(stream), z, n, target, coef, gscale, dz, partial);
    TG_CHECK_LAUNCH("bce_kernel");
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(64), 0, S(stream), partial, grid, 1.0 / (double)n, out1);
    TG_CHECK_LAUNCH("mean_final_kernel");
    return TG_OK;
}

// =================================================================================================
// Adam, axpby, lincomb, layout transposes
// =================================================================================================
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float one_minus_b1, float b2,
                                                   float one_minus_b2, float step_size, float bc2_sqrt, float eps,
                                                   float grad_scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gr = g[i] * grad_scale;
        const float mi = m[i] + one_minus_b1 * (gr - m[i]);       // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + one_minus_b2 * gr * gr;      // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;           // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - step_size * (mi / denom);
    }
}
// Scalars are taken as doubles and rounded to fp32 exactly where torch.optim.Adam rounds its Python floats
// (lerp weight 1-beta1, mul_ by beta2, addcmul_ value 1-beta2, addcdiv_ value -lr/bc1, division by sqrt(bc2)).
extern "C" int tg_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2, double eps,
                       int step, float grad_scale, tg_stream_t stream) {
    TG_REQUIRE(p && g && m && v && n > 0 && step >= 1, "tg_adam: bad arguments");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), p, g, m, v, n, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)(lr / bc1), (float)sqrt(bc2), (float)eps, grad_scale);
    TG_CHECK_LAUNCH("adam_kernel");
    return TG_OK;
}

// multi-tensor Adam: one launch walks a device-resident table of (p, g, m, v, n) segments; work item w covers
// elements [chunk*chunk_elems, ...) of segment seg (both packed in `work`)
__device__ __forceinline__ void adam_multi_body(const TgAdamSeg* __restrict__ segs, const int32_t* __restrict__ work,
                                                int chunk_elems, float one_minus_b1, float b2, float one_minus_b2,
                                                float step_size, float bc2_sqrt, float eps, float grad_scale) {
    const int seg = work[2 * blockIdx.x], chunk = work[2 * blockIdx.x + 1];
    const TgAdamSeg sg = segs[seg];
    const int64_t begin = (int64_t)chunk * chunk_elems;
    const int64_t end = min(sg.n, begin + chunk_elems);
    const bool al = ((reinterpret_cast<uintptr_t>(sg.p) | reinterpret_cast<uintptr_t>(sg.g) | reinterpret_cast<uintptr_t>(sg.m) |
                      reinterpret_cast<uintptr_t>(sg.v)) & 15) == 0;
    int64_t i = begin;
    if (al) {       // chunk_elems is a multiple of 4: 16-byte accesses over the aligned body
        const int64_t end4 = begin + ((end - begin) & ~(int64_t)3);
        for (i = begin + 4 * threadIdx.x; i < end4; i += 4 * 256) {
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(sg.g + i);
            f32x4 m4 = *reinterpret_cast<f32x4*>(sg.m + i), v4 = *reinterpret_cast<f32x4*>(sg.v + i), p4 = *reinterpret_cast<f32x4*>(sg.p + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gr = g4[e] * grad_scale;
                m4[e] = m4[e] + one_minus_b1 * (gr - m4[e]);
                v4[e] = v4[e] * b2 + one_minus_b2 * gr * gr;
                p4[e] = p4[e] - step_size * (m4[e] / (sqrtf(v4[e]) / bc2_sqrt + eps));
            }
            *reinterpret_cast<f32x4*>(sg.m + i) = m4;
            *reinterpret_cast<f32x4*>(sg.v + i) = v4;
            *reinterpret_cast<f32x4*>(sg.p + i) = p4;
        }
        i = end4;
    }
    for (i += threadIdx.x; i < end; i += 256) {
        const float gr = sg.g[i] * grad_scale;
        const float mi = sg.m[i] + one_minus_b1 * (gr - sg.m[i]);
        const float vi = sg.v[i] * b2 + one_minus_b2 * gr * gr;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        sg.m[i] = mi;
        sg.v[i] = vi;
        sg.p[i] = sg.p[i] - step_size * (mi / denom);
    }
}
__global__ __launch_bounds__(256) void adam_multi_kernel(const TgAdamSeg* __restrict__ segs, const int32_t* __restrict__ work,
                                                         int chunk_elems, float one_minus_b1, float b2, float one_minus_b2,
                                                         float step_size, float bc2_sqrt, float eps, float grad_scale) {
    adam_multi_body(segs, work, chunk_elems, one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, grad_scale);
}
// the two per-step scalars (step_size = lr / bias_correction1, sqrt(bias_correction2)) read from device memory: a launch
// captured in a hipGraph stays valid step after step, the host rewrites the two floats before each replay
__global__ __launch_bounds__(256) void adam_multi_s_kernel(const TgAdamSeg* __restrict__ segs, const int32_t* __restrict__ work,
                                                           int chunk_elems, float one_minus_b1, float b2, float one_minus_b2,
                                                           const float* __restrict__ scal, float eps, float grad_scale) {
    adam_multi_body(segs, work, chunk_elems, one_minus_b1, b2, one_minus_b2, scal[0], scal[1], eps, grad_scale);
}
struct TgFloats16 { float v[16]; };
__global__ void write_floats_kernel(float* __restrict__ dst, int n, TgFloats16 vals) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}
// dst[0..n) = vals_host[0..n), n <= 16, the values travelling as KERNEL ARGUMENTS (copied at launch): unlike an asynchronous
// copy from pinned memory, the host buffer may be rewritten as soon as this call returns
extern "C" int tg_write_floats(float* dst_dev, int n, const float* vals_host, tg_stream_t stream) {
    TG_REQUIRE(dst_dev && vals_host && n >= 1 && n <= 16, "tg_write_floats: bad arguments");
    TgFloats16 v = {};
    for (int i = 0; i < n; ++i) v.v[i] = vals_host[i];
    hipLaunchKernelGGL(write_floats_kernel, dim3(1), dim3(64), 0, S(stream), dst_dev, n, v);
    TG_CHECK_LAUNCH("write_floats_kernel");
    return TG_OK;
}
extern "C" int tg_adam_scalars(double lr, double beta1, double beta2, int step, float* out2_host) {
    TG_REQUIRE(out2_host && step >= 1, "tg_adam_scalars: bad arguments");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    out2_host[0] = (float)(lr / bc1);
    out2_host[1] = (float)sqrt(bc2);
    return TG_OK;
}
extern "C" int tg_adam_multi_s(const TgAdamSeg* segs_dev, const int32_t* work_dev, int nwork, int chunk_elems, double beta1,
                               double beta2, double eps, const float* scal_dev, float grad_scale, tg_stream_t stream) {
    TG_REQUIRE(segs_dev && work_dev && scal_dev && nwork > 0 && chunk_elems > 0, "tg_adam_multi_s: bad arguments");
    hipLaunchKernelGGL(adam_multi_s_kernel, dim3(nwork), dim3(256), 0, S(stream), segs_dev, work_dev, chunk_elems, (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), scal_dev, (float)eps, grad_scale);
    TG_CHECK_LAUNCH("adam_multi_s_kernel");
    return TG_OK;
}
extern "C" int tg_adam_multi(const TgAdamSeg* segs_dev, const int32_t* work_dev, int nwork, int chunk_elems, double lr, double beta1,
                             double beta2, double eps, int step, float grad_scale, tg_stream_t stream) {
    TG_REQUIRE(segs_dev && work_dev && nwork > 0 && chunk_elems > 0 && step >= 1, "tg_adam_multi: bad arguments");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nwork), dim3(256), 0, S(stream), segs_dev, work_dev, chunk_elems, (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)(lr / bc1), (float)sqrt(bc2), (float)eps, grad_scale);
    TG_CHECK_LAUNCH("adam_multi_kernel");
    return TG_OK;
}

__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, float a, float b, float* __restrict__ y,
                                                    int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = b == 0.f ? a * x[i] : a * x[i] + b * y[i];
}
extern "C" int tg_axpby(const float* x, float a, float b, float* y, int64_t n, tg_stream_t stream) {
    TG_REQUIRE(x && y && n > 0, "tg_axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), x, a, b, y, n);
    TG_CHECK_LAUNCH("axpby_kernel");
    return TG_OK;
}
__global__ __launch_bounds__(256) void lincomb_kernel(const float* __restrict__ x, float a, const float* __restrict__ y, float b,
                                                      float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = a * x[i] + b * y[i];
}
extern "C" int tg_lincomb(const float* x, float a, const float* y, float b, float* out, int64_t n, tg_stream_t stream) {
    TG_REQUIRE(x && y && out && n > 0, "tg_lincomb: bad arguments");
    hipLaunchKernelGGL(lincomb_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), x, a, y, b, out, n);
    TG_CHECK_LAUNCH("lincomb_kernel");
    return TG_OK;
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                  int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] * b[i];
}
extern "C" int tg_mul(const float* a, const float* b, float* out, int64_t n, tg_stream_t stream) {
    TG_REQUIRE(a && b && out && n > 0, "tg_mul: bad arguments");
    hipLaunchKernelGGL(mul_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), a, b, out, n);
    TG_CHECK_LAUNCH("mul_kernel");
    return TG_OK;
}

// out = a*b and keep = a from ONE read of a: the train step needs real*mask (train.py:181) and, stacked behind the generated
// batch, real itself (the loss trunk and the discriminator see [gen; real]) -- no separate copy, no concatenation
__global__ __launch_bounds__(256) void mul_keep_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                       float* __restrict__ keep, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = a[i];
        out[i] = v * b[i];
        keep[i] = v;
    }
}
extern "C" int tg_mul_keep(const float* a, const float* b, float* out, float* a_copy, int64_t n, tg_stream_t stream) {
    TG_REQUIRE(a && b && out && a_copy && n > 0, "tg_mul_keep: bad arguments");
    hipLaunchKernelGGL(mul_keep_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, S(stream), a, b, out, a_copy, n);
    TG_CHECK_LAUNCH("mul_keep_kernel");
    return TG_OK;
}

__global__ __launch_bounds__(256) void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                double n, int C, float eps, float momentum,
                                                                float* __restrict__ rm, float* __restrict__ rv,
                                                                int64_t* __restrict__ nbt) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && nbt) *nbt += 1;
    if (c >= C) return;
    const double r = (double)rstd[c];
    double var = 1.0 / (r * r) - (double)eps;
    if (var < 0.0) var = 0.0;
    rm[c] = (float)((1.0 - (double)momentum) * (double)rm[c] + (double)momentum * (double)mean[c]);
    rv[c] = (float)((1.0 - (double)momentum) * (double)rv[c] + (double)momentum * var * (n / (n - 1.0)));
}
extern "C" int tg_bn_running_update(const float* save_mean, const float* save_rstd, int64_t rows, int C, float eps,
                                    float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                    tg_stream_t stream) {
    TG_REQUIRE(save_mean && save_rstd && running_mean && running_var && rows > 1 && C > 0, "tg_bn_running_update: bad arguments");
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S(stream), save_mean, save_rstd, (double)rows,
                       C, eps, momentum, running_mean, running_var, num_batches_tracked);
    TG_CHECK_LAUNCH("bn_running_update_kernel");
    return TG_OK;
}

// ---- BatchNorm over `groups` passes stacked along the rows (statistics per pass) in ONE set of launches ---------------------------
// The train step runs D(fake) and D(real) as one stacked forward and their backward passes as one (train.py:202,211-218): the
// convolutions see 2B images, BatchNorm must see each pass by itself.  Per layer that was 2 x (reduce, finalise, apply) forward and
// 2 x (reduce, finalise, apply) + 3 accumulations backward; here: one reduce (groups side by side), one finalise, one apply --
// 6 launches instead of 15, bit-identical (same block geometry and summation order per group; parameter gradients summed in pass order).
__global__ __launch_bounds__(256) void bn_finalize_grouped_kernel(const float* __restrict__ partial, int nblocks_g, int groups, int C,
                                                                  const float* __restrict__ y, int64_t rows_g, float eps,
                                                                  float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    for (int g = 0; g < groups; ++g) {
        double q[2];
        int c;
        const bool own = final_reduce<2>(partial + (size_t)g * nblocks_g * 2 * C, nblocks_g, C, q, &c);
        if (own) {
            const double n = (double)rows_g;
            const double md = q[0] / n;
            double var = q[1] / n - md * md;
            if (var < 0.0) var = 0.0;
            mean_out[g * C + c] = (float)((double)y[(size_t)g * rows_g * C + c] + md);
            rstd_out[g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
        }
        __syncthreads();                       // final_reduce's LDS is reused by the next group
    }
}
__global__ __launch_bounds__(256) void bn_act_fwd4_grouped_kernel(const float* __restrict__ y, int64_t rows_g, int C, int qpp, int rlanes,
                                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  int act, float slope, float* __restrict__ out) {
    const int cq = threadIdx.x % qpp, rl = threadIdx.x / qpp;
    if (rl >= rlanes) return;
    const int g = blockIdx.z, nquads = C >> 2;
    const float* yg = y + (size_t)g * rows_g * C;
    float* og = out + (size_t)g * rows_g * C;
    for (int q0 = blockIdx.y * qpp; q0 < nquads; q0 += gridDim.y * qpp) {
        const int c0 = (q0 + cq) * 4;
        if (c0 >= C) continue;
        const f32x4 mv = *reinterpret_cast<const f32x4*>(mean + g * C + c0), rv = *reinterpret_cast<const f32x4*>(rstd + g * C + c0);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + c0), bv = *reinterpret_cast<const f32x4*>(beta + c0);
        for (int64_t r = (int64_t)blockIdx.x * rlanes + rl; r < rows_g; r += (int64_t)gridDim.x * rlanes) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(yg + r * C + c0);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = apply_act(bn_affine(v[e], mv[e], rv[e], gv[e], bv[e]), act, slope);
            *reinterpret_cast<f32x4*>(og + r * C + c0) = o;
        }
    }
}
extern "C" size_t tg_bn_grouped_ws_bytes(int64_t rows_per_group, int groups, int C) {
    if (rows_per_group <= 0 || groups <= 0 || C <= 0 || C % 4) return 0;
    const ColGeom4 g = col_geom4(rows_per_group, C);
    // partials of every group + (backward) the per-group dgamma / dbeta the apply pass reads
    return (align_up((size_t)groups * g.grid * 5 * C, 64) + align_up((size_t)groups * 2 * C, 64)) * sizeof(float);
}
extern "C" int tg_bn_fwd_grouped(const float* y, int64_t rows_per_group, int groups, int C, float eps, const float* gamma,
                                 const float* beta, int act, float slope, float* save_mean, float* save_rstd, float* out,
                                 float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(y && gamma && beta && save_mean && save_rstd && out && ws, "tg_bn_fwd_grouped: null pointer");
    TG_REQUIRE(groups >= 1 && groups <= 8 && C >= 4 && C <= 1024 && C % 4 == 0, "tg_bn_fwd_grouped: groups=%d C=%d unsupported", groups, C);
    TG_REQUIRE(rows_per_group > 1, "tg_bn_fwd_grouped: Expected more than 1 value per channel when training");
    TG_REQUIRE(ws_bytes >= tg_bn_grouped_ws_bytes(rows_per_group, groups, C), "tg_bn_fwd_grouped: workspace too small");
    const ColGeom4 g = col_geom4(rows_per_group, C);
    BnStatF f{y, C};
    hipLaunchKernelGGL((colreduce4_kernel<2, BnStatF>), dim3(g.grid * groups), dim3(256), 0, S(stream), f, rows_per_group, C, g.qpp,
                       g.rlanes, g.rows_per_block, ws, g.grid);
    TG_CHECK_LAUNCH("bn_stats_grouped");
    hipLaunchKernelGGL(bn_finalize_grouped_kernel, dim3(fr_grid(C)), dim3(256), 0, S(stream), ws, g.grid, groups, C, y, rows_per_group,
                       eps, save_mean, save_rstd);
    TG_CHECK_LAUNCH("bn_finalize_grouped");
    RowGeom rg = row_geom(rows_per_group, C);
    rg.grid.z = groups;
    hipLaunchKernelGGL(bn_act_fwd4_grouped_kernel, rg.grid, dim3(256), 0, S(stream), y, rows_per_group, C, rg.qpp, rg.rlanes, save_mean,
                       save_rstd, gamma, beta, act, slope, out);
    TG_CHECK_LAUNCH("bn_act_fwd4_grouped");
    return TG_OK;
}

// backward: per-group sums -> per-group dgamma_g / dbeta_g (what the apply pass of that group needs) in `gws`, parameter gradients
// = sum over the groups in pass order (fp32, exactly what accumulating the separate passes' results gave)
__global__ __launch_bounds__(256) void bn_bwd_final_grouped_kernel(const float* __restrict__ partial, int nblocks_g, int groups, int C,
                                                                   double n, const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                                   float* __restrict__ gws, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, float* __restrict__ dbias) {
    float sg = 0.f, sb = 0.f, sd = 0.f;
    int c = 0;
    bool own = false;
    for (int g = 0; g < groups; ++g) {
        double q[5];
        own = final_reduce<5>(partial + (size_t)g * nblocks_g * 5 * C, nblocks_g, C, q, &c);
        if (own) {
            const float db_ = (float)q[0], dg_ = (float)q[1];
            gws[(g * 2 + 0) * C + c] = dg_;
            gws[(g * 2 + 1) * C + c] = db_;
            const float dbi = (float)((double)gamma[c] * (double)rstd[g * C + c] * (q[2] - q[0] / n * q[4] - q[1] / n * q[3]));
            sg = g == 0 ? dg_ : dg_ + sg;          // axpby(x = this pass, y = running sum): 1*x + 1*y
            sb = g == 0 ? db_ : db_ + sb;
            sd = g == 0 ? dbi : dbi + sd;
        }
        __syncthreads();
    }
    if (own) {
        dgamma[c] = sg;
        dbeta[c] = sb;
        if (dbias) dbias[c] = sd;
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply4_grouped_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                                    int64_t rows_g, int C, int qpp, int rlanes,
                                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    int act, float slope, const float* __restrict__ gws,
                                                                    float* __restrict__ dy) {
    const int cq = threadIdx.x % qpp, rl = threadIdx.x / qpp;
    if (rl >= rlanes) return;
    const int g = blockIdx.z;
    const size_t o = (size_t)g * rows_g * C;
    const float inv_n = 1.0f / (float)rows_g;
    const int nquads = C >> 2;
    for (int q0 = blockIdx.y * qpp; q0 < nquads; q0 += gridDim.y * qpp) {
        const int c0 = (q0 + cq) * 4;
        if (c0 >= C) continue;
        const f32x4 mv = *reinterpret_cast<const f32x4*>(mean + g * C + c0), rv = *reinterpret_cast<const f32x4*>(rstd + g * C + c0);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + c0), bv = *reinterpret_cast<const f32x4*>(beta + c0);
        const f32x4 dgv = *reinterpret_cast<const f32x4*>(gws + (g * 2 + 0) * C + c0), dbv = *reinterpret_cast<const f32x4*>(gws + (g * 2 + 1) * C + c0);
        for (int64_t r = (int64_t)blockIdx.x * rlanes + rl; r < rows_g; r += (int64_t)gridDim.x * rlanes) {
            const f32x4 yv = *reinterpret_cast<const f32x4*>(y + o + r * C + c0);
            const f32x4 dv = *reinterpret_cast<const f32x4*>(dout + o + r * C + c0);
            f32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xh = (yv[e] - mv[e]) * rv[e];
                float gg = dv[e] * act_grad(xh * gv[e] + bv[e], act, slope);
                ov[e] = gv[e] * rv[e] * (gg - dbv[e] * inv_n - xh * dgv[e] * inv_n) * 1.f;
            }
            *reinterpret_cast<f32x4*>(dy + o + r * C + c0) = ov;
        }
    }
}
extern "C" int tg_bn_act_bwd_grouped(const float* dout, const float* y, int64_t rows_per_group, int groups, int C, const float* mean,
                                     const float* rstd, const float* gamma, const float* beta, int act, float slope, float* dy,
                                     float* dgamma, float* dbeta, float* dbias, float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(dout && y && mean && rstd && gamma && beta && dy && dgamma && dbeta && ws, "tg_bn_act_bwd_grouped: null pointer");
    TG_REQUIRE(groups >= 1 && groups <= 8 && C >= 4 && C <= 1024 && C % 4 == 0 && rows_per_group > 0, "tg_bn_act_bwd_grouped: bad dims");
    TG_REQUIRE(ws_bytes >= tg_bn_grouped_ws_bytes(rows_per_group, groups, C), "tg_bn_act_bwd_grouped: workspace too small");
    const ColGeom4 g = col_geom4(rows_per_group, C);
    float* gws = ws + align_up((size_t)groups * g.grid * 5 * C, 64);
    BnBwdF f{dout, y, mean, rstd, gamma, beta, C, act, slope, nullptr};
    hipLaunchKernelGGL((colreduce4_kernel<5, BnBwdF>), dim3(g.grid * groups), dim3(256), 0, S(stream), f, rows_per_group, C, g.qpp,
                       g.rlanes, g.rows_per_block, ws, g.grid);
    TG_CHECK_LAUNCH("bn_bwd_reduce_grouped");
    hipLaunchKernelGGL(bn_bwd_final_grouped_kernel, dim3(fr_grid(C)), dim3(256), 0, S(stream), ws, g.grid, groups, C, (double)rows_per_group,
                       gamma, rstd, gws, dgamma, dbeta, dbias);
    TG_CHECK_LAUNCH("bn_bwd_final_grouped");
    RowGeom rg = row_geom(rows_per_group, C);
    rg.grid.z = groups;
    hipLaunchKernelGGL(bn_bwd_apply4_grouped_kernel, rg.grid, dim3(256), 0, S(stream), dout, y, rows_per_group, C, rg.qpp, rg.rlanes, mean,
                       rstd, gamma, beta, act, slope, gws, dy);
    TG_CHECK_LAUNCH("bn_bwd_apply4_grouped");
    return TG_OK;
}

// running-statistics updates of several passes, applied one after the other in `order` (indices into the per-group statistics) by
// ONE launch: the reference's D(fake), D(real), D(fake.detach()) update model.N's buffers in that order (train.py:202,211,212)
struct TgBnOrder { int n; int idx[8]; };
__global__ __launch_bounds__(256) void bn_running_update_multi_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                      double n, int C, float eps, float momentum, TgBnOrder ord,
                                                                      float* __restrict__ rm, float* __restrict__ rv,
                                                                      int64_t* __restrict__ nbt) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && nbt) *nbt += ord.n;
    if (c >= C) return;
    float m_ = rm[c], v_ = rv[c];
    for (int k = 0; k < ord.n; ++k) {
        const int g = ord.idx[k];
        const double r = (double)rstd[g * C + c];
        double var = 1.0 / (r * r) - (double)eps;
        if (var < 0.0) var = 0.0;
        m_ = (float)((1.0 - (double)momentum) * (double)m_ + (double)momentum * (double)mean[g * C + c]);
        v_ = (float)((1.0 - (double)momentum) * (double)v_ + (double)momentum * var * (n / (n - 1.0)));
    }
    rm[c] = m_;
    rv[c] = v_;
}
extern "C" int tg_bn_running_update_multi(const float* save_mean, const float* save_rstd, int64_t rows_per_group, int C, float eps,
                                          float momentum, const int* order, int norder, float* running_mean, float* running_var,
                                          int64_t* num_batches_tracked, tg_stream_t stream) {
    TG_REQUIRE(save_mean && save_rstd && running_mean && running_var && order && rows_per_group > 1 && C > 0 && norder >= 1 && norder <= 8,
               "tg_bn_running_update_multi: bad arguments");
    TgBnOrder ord = {};
    ord.n = norder;
    for (int k = 0; k < norder; ++k) {
        TG_REQUIRE(order[k] >= 0 && order[k] < 8, "tg_bn_running_update_multi: order[%d] = %d", k, order[k]);
        ord.idx[k] = order[k];
    }
    hipLaunchKernelGGL(bn_running_update_multi_kernel, dim3(cdiv(C, 256)), dim3(256), 0, S(stream), save_mean, save_rstd,
                       (double)rows_per_group, C, eps, momentum, ord, running_mean, running_var, num_batches_tracked);
    TG_CHECK_LAUNCH("bn_running_update_multi_kernel");
    return TG_OK;
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int B, int C, int HW,
                                                           float* __restrict__ y) {
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t t = i / C;
        const int64_t hw = t % HW, b = t / HW;
        y[i] = x[(b * C + c) * HW + hw];
    }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, int B, int C, int HW,
                                                           float* __restrict__ y) {
    const int64_t total = (int64_t)B * C * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t hw = i % HW;
        const int64_t t = i / HW;
        const int c = (int)(t % C);
        const int64_t b = t / C;
        y[i] = x[(b * HW + hw) * C + c];
    }
}
extern "C" int tg_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float* y, tg_stream_t stream) {
    TG_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "tg_nchw_to_nhwc: bad arguments");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid((int64_t)B * C * H * W, 256)), dim3(256), 0, S(stream), x, B, C, H * W, y);
    TG_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return TG_OK;
}
extern "C" int tg_nhwc_to_nchw(const float* x, int B, int C, int H, int W, float* y, tg_stream_t stream) {
    TG_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "tg_nhwc_to_nchw: bad arguments");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid((int64_t)B * C * H * W, 256)), dim3(256), 0, S(stream), x, B, C, H * W, y);
    TG_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return TG_OK;
}
