// Bandwidth-bound convolutions with a single input or a single output channel (enc1, D conv0, VGG conv1_1,
// `final`, D conv4 and their gradients).  As GEMMs they have N or K of 1 and would use <5 % of an MFMA tile.
// Here a wavefront's 64 lanes ARE 64 channels of the wide side, so every global access to the wide tensor is
// one coalesced 256-byte row.
//   * 1 -> N channels (c1conv / c1wgrad): the 1-channel source patch of a 16x16 output tile is staged in LDS
//     once (zero halo, mask pre-multiplied), so the tap loop is branch-free: one broadcast ds_read + one FMA.
//   * C -> 1 channel (to1conv / to1wgrad): tap loops are compile-time unrolled and predicated (clamped address,
//     0/1 factor) so all tap loads of 4 pixels are in flight together; the reduction over channels shares one
//     7-shuffle butterfly between 4 pixels.
#include <stdlib.h>

#include "igemm_params.h"

__device__ __forceinline__ int weight_tap(const IGemmParams& p, int ty, int tx) {
    return (p.ky0 + ty * p.kstep) * p.KW + (p.kx0 + tx * p.kstep);
}
__device__ __forceinline__ size_t out_pixel(const IGemmParams& p, int b, int oy, int ox) {
    return ((size_t)b * p.DH + (oy * p.ds + p.dy0)) * p.DW + (ox * p.ds + p.dx0);
}

constexpr int C1_T = 16;   // output tile edge of the 1-channel-source kernels

struct C1Geom {
    int tiles_x, tiles_y, PH, PW, sy_min, sx_min;
};

// stage the [PH][PW] source patch of tile (b, oy0, ox0): zero outside the image, mask pre-multiplied
__device__ __forceinline__ void c1_stage_patch(const float* __restrict__ src, const float* __restrict__ amask, float* patch,
                                               int b, int py0, int px0, int PH, int PW, int IH, int IW) {
    // Six elements per thread and trip (a 37 x 37 patch -- 7x7 stride 2 -- is one trip), loads unconditional at clamped
    // addresses and all in flight together: the rolled, branchy form (load, wait, store per element) serialised up to six memory
    // round trips per tile, ~10 us of the 18 us a tile of the persistent weight-gradient kernel took.
    constexpr int U = 6;
    const int n = PH * PW;
    for (int i0 = threadIdx.x; i0 < n; i0 += 256 * U) {
        float v[U], m[U];
        bool ok[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = i0 + 256 * k;
            const int ic = i < n ? i : 0;
            const int py = ic / PW, px = ic - py * PW;
            const int iy = py0 + py, ix = px0 + px;
            ok[k] = i < n && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
            const size_t pix = ok[k] ? ((size_t)b * IH + iy) * IW + ix : 0;
            v[k] = src[pix];
            m[k] = amask ? amask[pix] : 1.f;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int i = i0 + 256 * k;
            float o = v[k];
            if (amask) o *= m[k];
            if (i < n) patch[i] = ok[k] ? o : 0.f;
        }
    }
}

// ---- 1 source channel -> N channels (N % 64 == 0): lane = output channel ----------------------------------
template <int TH_, int TW_>   // taps (0,0 = runtime)
__global__ __launch_bounds__(256) void c1conv_kernel(const IGemmParams p, const C1Geom q) {
    extern __shared__ float sm[];
    float* patch = sm;                 // [PH*PW]
    float* wl = sm + q.PH * q.PW;      // [taps][64] (runtime-tap variant only)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + lane;
    const int TH = TH_ ? TH_ : p.TH, TW = TW_ ? TW_ : p.TW;
    int tile = blockIdx.x;
    const int txi = tile % q.tiles_x;
    tile /= q.tiles_x;
    const int tyi = tile % q.tiles_y, b = tile / q.tiles_y;
    const int oy0 = tyi * C1_T, ox0 = txi * C1_T;
    c1_stage_patch(p.src, p.amask, patch, b, oy0 * p.ss + q.sy_min, ox0 * p.ss + q.sx_min, q.PH, q.PW, p.IH, p.IW);
    constexpr int NT = TH_ * TW_ > 0 ? TH_ * TW_ : 1;
    float w[NT];
    if constexpr (TH_ > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) w[t] = p.wmat[(size_t)n * p.Kfull + weight_tap(p, t / TW_, t % TW_)];
    } else {
        for (int i = threadIdx.x; i < TH * TW * 64; i += 256)
            wl[i] = p.wmat[(size_t)(blockIdx.y * 64 + (i & 63)) * p.Kfull + weight_tap(p, (i >> 6) / TW, (i >> 6) % TW)];
    }
    __syncthreads();
    const float bias = p.bias ? p.bias[n] : 0.f;
    const int oyy = p.sy0 - q.sy_min, oxx = p.sx0 - q.sx_min;
    // each wave owns 4 rows of the 16x16 tile
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const int ty_o = wave * 4 + r;
        const int oy = oy0 + ty_o;
        if (oy >= p.OH) break;
#pragma unroll 4
        for (int tx_o = 0; tx_o < C1_T; ++tx_o) {
            const int ox = ox0 + tx_o;
            if (ox >= p.OW) break;
            const float* pb = patch + (ty_o * p.ss + oyy) * q.PW + tx_o * p.ss + oxx;
            float acc = 0.f;
            if constexpr (TH_ > 0) {
#pragma unroll
                for (int ty = 0; ty < TH_; ++ty)
#pragma unroll
                    for (int tx = 0; tx < TW_; ++tx) acc = fmaf(pb[ty * p.tstep * q.PW + tx * p.tstep], w[ty * TW_ + tx], acc);
            } else {
                for (int ty = 0; ty < TH; ++ty)
                    for (int tx = 0; tx < TW; ++tx)
                        acc = fmaf(pb[ty * p.tstep * q.PW + tx * p.tstep], wl[(ty * TW + tx) * 64 + lane], acc);
            }
            const size_t opix = out_pixel(p, b, oy, ox);
            float v = acc + bias;
            if (p.rowscale) v *= p.rowscale[opix];
            v = apply_act(v, p.act, p.slope);
            if (p.gate) v *= gate_factor(p, opix * p.N + n);
            float* d = p.dst + opix * p.N + n;
            if (p.accumulate) v += *d;
            *d = v;
        }
    }
}


// ---- 1 source channel -> N channels on the fp32 MFMA (K = taps) ---------------------------------------------------------
// The lane-per-channel kernel above issues ONE broadcast ds_read_b32 per FMA: for enc1 (7x7: 49 taps) that is 12.8 M LDS
// wave-instructions, and the kernel runs at 0.93 TB/s (12 % of HBM) although it only has to write 67 MB.  As a GEMM the layer
// is M = pixels, N = 64, K = 49: thin, but v_mfma_f32_32x32x2_f32 needs just 25 K-steps, i.e. the matrix pipe is busy for
// ~11 us chip-wide -- the same as the output stream at the HBM rate.  A wave owns 4 rows x 16 pixels x 64 channels of the
// 16x16 tile (2 x 2 accumulator tiles); its A operand is an im2col view of the 1-channel patch in LDS (lane = pixel, lane
// half = one of the step's two taps: one ds_read_b32 per MFMA pair), its B operand the zero-padded [K][64] weight image in
// LDS.  The epilogue bounces each 32x32 tile through LDS (tile_rows4) for 16-byte NHWC stores.  Exact fp32 arithmetic
// (the MFMA is an fma chain over k).
template <int TH_, int TW_>
__global__ __launch_bounds__(256) void c1mfma_kernel(const IGemmParams p, const C1Geom q) {
    constexpr int NT = TH_ * TW_, KS = (NT + 1) / 2;
    constexpr int WLP = 68;                      // row pitch of the weight image: the transposing stores spread over the banks
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* wl = sm;                              // [2*KS][WLP]  (row NT is zero when NT is odd)
    float* scratch = sm + 2 * KS * WLP;          // [4 waves][32*36]
    float* patch = scratch + 4 * 32 * 36;        // [PH*PW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int tile = blockIdx.x;
    const int txi = tile % q.tiles_x;
    tile /= q.tiles_x;
    const int tyi = tile % q.tiles_y, b = tile / q.tiles_y;
    const int oy0 = tyi * C1_T, ox0 = txi * C1_T;
    const int n0 = blockIdx.y * 64;
    {
        // Staging with every global load of the workgroup in flight at once: a rolled loop (runtime trip count) costs one
        // L2 / HBM round trip per iteration -- ~15 us of the first version's 54 us on enc1, where all workgroups of a CU
        // start together and nothing else covers it.
        constexpr int PMAX = (15 * 2 + TH_) * (15 * 2 + TW_);      // largest patch (stride 2)
        constexpr int PIT = (PMAX + 255) / 256, WIT = (NT * 64 + 255) / 256;
        const int py0 = oy0 * p.ss + q.sy_min, px0 = ox0 * p.ss + q.sx_min, pn = q.PH * q.PW;
        float pv[PIT], pm[PIT], wv[WIT];
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int i = threadIdx.x + 256 * it;
            const int py = i / q.PW, px = i - py * q.PW;
            const int iy = py0 + py, ix = px0 + px;
            const bool in = i < pn && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const size_t pix = in ? ((size_t)b * p.IH + iy) * p.IW + ix : 0;
            pv[it] = p.src[pix];
            pm[it] = !in ? 0.f : (p.amask ? p.amask[pix] : 1.f);
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {       // k fastest: a channel's taps are contiguous in memory
            const int i = threadIdx.x + 256 * it;
            const int c = i / NT, k = i - c * NT;
            wv[it] = i < NT * 64 ? p.wmat[(size_t)(n0 + c) * p.Kfull + weight_tap(p, k / TW_, k % TW_)] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < PIT; ++it) {
            const int i = threadIdx.x + 256 * it;
            if (i < pn) patch[i] = pm[it] != 0.f ? pv[it] * pm[it] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = threadIdx.x + 256 * it;
            const int c = i / NT, k = i - c * NT;
            if (i < NT * 64) wl[k * WLP + c] = wv[it];
        }
        if constexpr (NT & 1)
            if (threadIdx.x < 64) wl[NT * WLP + threadIdx.x] = 0.f;      // the padded K row
    }
    __syncthreads();
    const int h = lane >> 5, pi = lane & 31;
    const int oyy = p.sy0 - q.sy_min, oxx = p.sx0 - q.sx_min;
    // patch address of this lane's pixel in M tile m (rows 2m, 2m+1 of the wave's 4 rows)
    int abase[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
        abase[m] = ((wave * 4 + 2 * m + (pi >> 4)) * p.ss + oyy) * q.PW + (pi & 15) * p.ss + oxx;
    const int ty_step = p.tstep * q.PW, tx_step = p.tstep;
    float* sc = scratch + wave * (32 * 36);
    // epilogue operands are requested BEFORE the MFMAs they follow: a load issued between two output stores cannot be
    // hoisted by the compiler (the stores may alias it), and 16 load -> wait -> store round trips per wave cost more
    // than the whole K loop
    const int ecc = 4 * (lane & 7), err = lane >> 3;
    f32x4 bias4[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
        bias4[nt] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + 32 * nt + ecc) : f32x4{0.f, 0.f, 0.f, 0.f};
    // M tile by M tile: the (asynchronous) output stores of tile 0 drain underneath the MFMAs of tile 1, and the four
    // workgroups of a CU, which start together, stop marching through their load / MFMA / store phases in lockstep
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        size_t opix[4];
        float rs[4];
        bool ok[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {            // tile_rows4 hands this lane rows err + 8t of the 32-pixel tile
            const int rr = err + 8 * t;
            const int oy = oy0 + wave * 4 + 2 * m + (rr >> 4), ox = ox0 + (rr & 15);
            ok[t] = oy < p.OH && ox < p.OW;
            opix[t] = ok[t] ? out_pixel(p, b, oy, ox) : 0;
            rs[t] = p.rowscale ? p.rowscale[opix[t]] : 1.f;
        }
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // taps k0 = 2s (lane half 0) and k1 = 2s+1 (lane half 1; the zero weight row when k1 == NT, any valid address)
            const int k0 = 2 * s, k1 = 2 * s + 1 < NT ? 2 * s + 1 : 2 * s;
            const int off0 = (k0 / TW_) * ty_step + (k0 % TW_) * tx_step, off1 = (k1 / TW_) * ty_step + (k1 % TW_) * tx_step;
            float a = patch[abase[m] + (h ? off1 : off0)];
            if (2 * s + 1 >= NT && h) a = 0.f;                      // padded K row: 0 * w(=0), never 0 * inf
            const float b0 = wl[(2 * s + h) * WLP + pi], b1 = wl[(2 * s + h) * WLP + 32 + pi];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            // (tile_rows4 inlined: the row index t must be a compile-time constant to address opix / rs / ok)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[nt][r];
            const int n = n0 + 32 * nt + ecc;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 v = *reinterpret_cast<const f32x4*>(sc + (err + 8 * t) * 36 + ecc);
                if (!ok[t]) continue;
                v = (v + bias4[nt]) * rs[t];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act, p.slope);
                float* d = p.dst + opix[t] * p.N + n;
                if (p.gate) {
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(p.gate + opix[t] * p.N + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gv[e] > 0.f ? 1.f : (p.gate_act == TG_ACT_LEAKY ? p.gate_slope : 0.f);
                }
                if (p.accumulate) v += *reinterpret_cast<const f32x4*>(d);
                *reinterpret_cast<f32x4*>(d) = v;
            }
        }
    }
}

// ---- 64 channels -> 1 channel: lane = (pixel of a 4-pixel row segment, channel quad) --------------------------
// Each lane loads float4 (4 channels), so one wave instruction fetches the full 256-B rows of 4 neighbouring
// pixels; the row base is wave-uniform (scalar ALU), the tap loops are unrolled and predicated, and the channel
// reduction is a 4-step shuffle inside each 16-lane group.  Requires OW % 4 == 0.
template <int TH_, int TW_>
__device__ __forceinline__ void to1conv64_body(const IGemmParams& p) {
    constexpr int NT = TH_ * TW_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane >> 4, cq = lane & 15;
    f32x4 w[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        w[t] = *reinterpret_cast<const f32x4*>(p.wmat + (size_t)weight_tap(p, t / TW_, t % TW_) * 64 + 4 * cq);
    const float bias = p.bias ? p.bias[0] : 0.f;
    const int quads = p.M >> 2;
    const int waves_total = gridDim.x * 4;
    for (int q0 = blockIdx.x * 4 + wave; q0 < quads; q0 += waves_total) {
        const int q = __builtin_amdgcn_readfirstlane(q0);
        const int m0 = 4 * q;
        const int ox0 = m0 % p.OW;
        const int t2 = m0 / p.OW;
        const int oy = t2 % p.OH, b = t2 / p.OH;
        const int ox = ox0 + e;
        float acc = 0.f;
#pragma unroll
        for (int ty = 0; ty < TH_; ++ty) {
            const int iy = oy * p.ss + p.sy0 + ty * p.tstep;
            const bool vy = iy >= 0 && iy < p.IH;
            const size_t rowbase = ((size_t)b * p.IH + min(max(iy, 0), p.IH - 1)) * p.IW;
#pragma unroll
            for (int tx = 0; tx < TW_; ++tx) {
                const int ix = ox * p.ss + p.sx0 + tx * p.tstep;
                const bool vx = ix >= 0 && ix < p.IW;
                const size_t pix = rowbase + min(max(ix, 0), p.IW - 1);
                float f = (vy && vx) ? 1.f : 0.f;
                if (p.amask) f *= p.amask[pix];
                const f32x4 x = *reinterpret_cast<const f32x4*>(p.src + pix * 64 + 4 * cq);
                const f32x4 ww = w[ty * TW_ + tx];
                acc = fmaf(f, x[0] * ww[0] + x[1] * ww[1] + x[2] * ww[2] + x[3] * ww[3], acc);
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (cq == 0) {
            const size_t opix = out_pixel(p, b, oy, ox);
            float r = acc + bias;
            if (p.rowscale) r *= p.rowscale[opix];
            r = apply_act(r, p.act, p.slope);
            if (p.gate) r *= gate_factor(p, opix);
            if (p.accumulate) r += p.dst[opix];
            p.dst[opix] = r;
        }
    }
}
template <int TH_, int TW_>
__global__ __launch_bounds__(256) void to1conv64_kernel(const IGemmParams p) { to1conv64_body<TH_, TW_>(p); }
// the parity classes of a stride-2 dgrad as one launch: blockIdx.y = class; the classes walk the same source pixels at the
// same time, so dy comes from memory once and from L2 three times
__global__ __launch_bounds__(256) void to1conv64_multi22_kernel(const IGemmMulti pm) { to1conv64_body<2, 2>(pm.c[blockIdx.y]); }


// ---- 64 channels -> 1 channel, 3x3 stride 1, through an LDS patch ------------------------------------------------------------
// to1conv64_kernel fetches every tap straight from global memory: 9 x the tensor goes through the texture path (L1 hits, but
// 2.4 GB of TA traffic for `final`'s 268 MB: 2.2 TB/s algorithmic, 27 % of HBM).  Here a workgroup stages the (4+2) x (32+2)
// pixel x 64 channel patch of a 4 x 32 output tile ONCE (52 KB; zero outside the image, mask pre-multiplied) and the nine
// taps are 16-byte LDS reads of four neighbouring pixels = 1 KB contiguous per wave instruction (conflict-free).  Lane =
// (pixel of a 4-pixel row segment, channel quad) and the 16-lane shuffle reduction are those of to1conv64_kernel.
// 4 x 16 output pixels per tile: 28 KB of LDS, five workgroups per CU.  (4 x 32 -- 52 KB, three per CU -- has 6 % less halo traffic
// but too few loads in flight: `final` forward 87 -> 79 us, its weight gradient 77 -> 65 us with the narrower tile.)
constexpr int T1_TH = 4, T1_TW = 16, T1_PH = T1_TH + 2, T1_PW = T1_TW + 2;
// BN-on-load: a thread's staging slots all belong to ONE channel quad (256 % 16 == 0), its constants are fetched once
template <int NIT>
__device__ __forceinline__ void bn_in_stage(const BnIn& bn, f32x4 (&v)[NIT], int c0) {
    const f32x4 mv = *reinterpret_cast<const f32x4*>(bn.mean + c0), rv = *reinterpret_cast<const f32x4*>(bn.rstd + c0);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(bn.gamma + c0), bv = *reinterpret_cast<const f32x4*>(bn.beta + c0);
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[it][k] = apply_act(bn_affine(v[it][k], mv[k], rv[k], gv[k], bv[k]), bn.act, bn.slope);
}
// BNIN: the source is act(BN(src)) (IGemmParams::in_bn, no source mask): its own instantiation, with the in-range flags as a bit
// mask instead of seven floats -- the sixteen per-channel constants then fit without costing a resident wave.
template <bool BNIN>
__global__ __launch_bounds__(256) void to1conv64_lds_kernel(const IGemmParams p, int tiles_x, int tiles_y, int sy_min, int sx_min) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [T1_PH][T1_PW][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane >> 4, cq = lane & 15;
    int tile = blockIdx.x;
    const int txi = tile % tiles_x;
    tile /= tiles_x;
    const int tyi = tile % tiles_y, b = tile / tiles_y;
    const int oy0 = tyi * T1_TH, ox0 = txi * T1_TW;
    f32x4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
        w[t] = *reinterpret_cast<const f32x4*>(p.wmat + (size_t)weight_tap(p, t / 3, t % 3) * 64 + 4 * cq);
    // stage the patch: thread = (pixel slot, channel quad); all 13 loads of a thread in flight together
    constexpr int NSLOT = T1_PH * T1_PW * 16, NIT = (NSLOT + 255) / 256;
    f32x4 v[NIT];
    float f[BNIN ? 1 : NIT];
    uint32_t inbits = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int slot = threadIdx.x + 256 * it, pp = slot >> 4;
        const int py = pp / T1_PW, px = pp - py * T1_PW;
        const int iy = oy0 + sy_min + py, ix = ox0 + sx_min + px;
        const bool in = slot < NSLOT && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        const size_t pix = in ? ((size_t)b * p.IH + iy) * p.IW + ix : 0;
        v[it] = *reinterpret_cast<const f32x4*>(p.src + pix * 64 + 4 * (slot & 15));
        if constexpr (BNIN) inbits |= (in ? 1u : 0u) << it;
        else f[it] = !in ? 0.f : (p.amask ? p.amask[pix] : 1.f);
    }
    if constexpr (BNIN) bn_in_stage<NIT>(p.in_bn, v, 4 * (threadIdx.x & 15));
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int slot = threadIdx.x + 256 * it;
        f32x4 o;
        if constexpr (BNIN) {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (inbits >> it) & 1u ? v[it][k] : 0.f;
        } else {
            o = v[it] * f[it];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = f[it] != 0.f ? o[k] : 0.f;
        }
        if (slot < NSLOT) *reinterpret_cast<f32x4*>(sm + 4 * slot) = o;
    }
    __syncthreads();
    const float bias = p.bias ? p.bias[0] : 0.f;
    const int oyy = p.sy0 - sy_min, oxx = p.sx0 - sx_min;
    // wave w owns tile row w: 8 segments of 4 pixels
#pragma unroll 2
    for (int seg = 0; seg < T1_TW / 4; ++seg) {
        const int tx_o = 4 * seg + e;
        const float* pb = sm + (((wave + oyy) * T1_PW) + tx_o + oxx) * 64 + 4 * cq;
        float acc = 0.f;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(pb + ((ty * p.tstep) * T1_PW + tx * p.tstep) * 64);
                const f32x4 ww = w[ty * 3 + tx];
                acc += x[0] * ww[0] + x[1] * ww[1] + x[2] * ww[2] + x[3] * ww[3];
            }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        const int oy = oy0 + wave, ox = ox0 + tx_o;
        if (cq == 0 && oy < p.OH && ox < p.OW) {
            const size_t opix = out_pixel(p, b, oy, ox);
            float r = acc + bias;
            if (p.rowscale) r *= p.rowscale[opix];
            r = apply_act(r, p.act, p.slope);
            if (p.gate) r *= gate_factor(p, opix);
            if (p.accumulate) r += p.dst[opix];
            p.dst[opix] = r;
        }
    }
}


// ---- 64 channels -> 1 channel, the FOUR 2x2-tap parity classes of a 4x4 stride-2 dgrad (D conv0's input gradient) from ONE LDS
// patch.  to1conv64_multi22_kernel fetches every tap of every class straight from global memory: each dy pixel goes through the
// texture path sixteen times (1 GB for 67 MB of dy: 60 us, 1.1 TB/s algorithmic).  Here a workgroup stages the (4+2) x (16+2) pixel
// x 64 channel patch of a 4 x 16 tile of the class grid once (as to1conv64_lds_kernel does) and all four classes read their
// taps from it; class c writes dst pixel (oy * ds + dy0_c, ox * ds + dx0_c).
__global__ __launch_bounds__(256) void to1conv64_multi22_lds_kernel(const IGemmMulti pm, int tiles_x, int tiles_y, int sy_min, int sx_min) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [T1_PH][T1_PW][64]
    const IGemmParams& p0 = pm.c[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane >> 4, cq = lane & 15;
    int tile = blockIdx.x;
    const int txi = tile % tiles_x;
    tile /= tiles_x;
    const int tyi = tile % tiles_y, b = tile / tiles_y;
    const int oy0 = tyi * T1_TH, ox0 = txi * T1_TW;
    constexpr int NSLOT = T1_PH * T1_PW * 16, NIT = (NSLOT + 255) / 256;
    f32x4 v[NIT];
    uint32_t inbits = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int slot = threadIdx.x + 256 * it, pp = slot >> 4;
        const int py = pp / T1_PW, px = pp - py * T1_PW;
        const int iy = oy0 + sy_min + py, ix = ox0 + sx_min + px;
        const bool in = slot < NSLOT && iy >= 0 && iy < p0.IH && ix >= 0 && ix < p0.IW;
        const size_t pix = in ? ((size_t)b * p0.IH + iy) * p0.IW + ix : 0;
        v[it] = *reinterpret_cast<const f32x4*>(p0.src + pix * 64 + 4 * (slot & 15));
        inbits |= (in ? 1u : 0u) << it;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int slot = threadIdx.x + 256 * it;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (inbits >> it) & 1u ? v[it][k] : 0.f;
        if (slot < NSLOT) *reinterpret_cast<f32x4*>(sm + 4 * slot) = o;
    }
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        const IGemmParams& p = pm.c[c];
        f32x4 w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            w[t] = *reinterpret_cast<const f32x4*>(p.wmat + (size_t)weight_tap(p, t >> 1, t & 1) * 64 + 4 * cq);
        const float bias = p.bias ? p.bias[0] : 0.f;
        const int oyy = p.sy0 - sy_min, oxx = p.sx0 - sx_min;
#pragma unroll 2
        for (int seg = 0; seg < T1_TW / 4; ++seg) {
            const int tx_o = 4 * seg + e;
            const float* pb = sm + (((wave + oyy) * T1_PW) + tx_o + oxx) * 64 + 4 * cq;
            float acc = 0.f;
#pragma unroll
            for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                for (int tx = 0; tx < 2; ++tx) {
                    const f32x4 x = *reinterpret_cast<const f32x4*>(pb + ((ty * p.tstep) * T1_PW + tx * p.tstep) * 64);
                    const f32x4 ww = w[ty * 2 + tx];
                    acc += x[0] * ww[0] + x[1] * ww[1] + x[2] * ww[2] + x[3] * ww[3];
                }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
            const int oy = oy0 + wave, ox = ox0 + tx_o;
            if (cq == 0 && oy < p.OH && ox < p.OW) {
                const size_t opix = out_pixel(p, b, oy, ox);
                float r = acc + bias;
                if (p.rowscale) r *= p.rowscale[opix];
                r = apply_act(r, p.act, p.slope);
                if (p.gate) r *= gate_factor(p, opix);
                if (p.accumulate) r += p.dst[opix];
                p.dst[opix] = r;
            }
        }
    }
}

// ---- C channels -> 1 channel for WIDE C (C % 256 == 0: the discriminator's last conv, 512 -> 1, 4x4) ---------------------------
// A wave owns one output pixel at a time: lane = channel quad j*64 + lane of every 256-channel group, so a tap of one pixel is
// C/256 coalesced 1 KB rows; the lane's weights (taps x C/256 quads) live in registers for the whole launch and the channel
// reduction is one 6-step butterfly per output.  (On the MFMA kernel this layer ran at 1.6 TF: N = 1 wastes 31/32 of a tile.)
template <int TH_, int TW_, int CQ>
__global__ __launch_bounds__(256) void to1convw_kernel(const IGemmParams p) {
    constexpr int NT = TH_ * TW_;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 w[NT][CQ];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < CQ; ++j)
            w[t][j] = *reinterpret_cast<const f32x4*>(p.wmat + (size_t)weight_tap(p, t / TW_, t % TW_) * p.C + 256 * j + 4 * lane);
    const float bias = p.bias ? p.bias[0] : 0.f;
    for (int m0 = blockIdx.x * 4 + wave; m0 < p.M; m0 += gridDim.x * 4) {
        const int m = __builtin_amdgcn_readfirstlane(m0);
        const int ox = m % p.OW, t2 = m / p.OW;
        const int oy = t2 % p.OH, b = t2 / p.OH;
        float acc = 0.f;
#pragma unroll
        for (int ty = 0; ty < TH_; ++ty) {
            const int iy = oy * p.ss + p.sy0 + ty * p.tstep;
            if (iy < 0 || iy >= p.IH) continue;                        // wave-uniform
            // the taps of a kernel row: loads unconditional at clamped columns and in flight together, a tap outside the image
            // is dropped by its factor (a branch per tap put each tap's loads in a basic block of their own: one round trip per
            // tap, sixteen in a row per output of the discriminator's last conv)
            f32x4 xr[TW_][CQ];
            float fr[TW_];
#pragma unroll
            for (int tx = 0; tx < TW_; ++tx) {
                const int ix = ox * p.ss + p.sx0 + tx * p.tstep;
                const bool okx = ix >= 0 && ix < p.IW;
                const size_t pix = ((size_t)b * p.IH + iy) * p.IW + (okx ? ix : 0);
                const float mk = p.amask ? p.amask[pix] : 1.f;
                fr[tx] = okx ? mk : 0.f;
#pragma unroll
                for (int j = 0; j < CQ; ++j) xr[tx][j] = *reinterpret_cast<const f32x4*>(p.src + pix * p.C + 256 * j + 4 * lane);
            }
#pragma unroll
            for (int tx = 0; tx < TW_; ++tx) {
                const int ix = ox * p.ss + p.sx0 + tx * p.tstep;
                if (ix < 0 || ix >= p.IW) continue;                    // (wave-uniform; keeps the summation order of the taps that exist)
                float part = 0.f;
#pragma unroll
                for (int j = 0; j < CQ; ++j) {
                    const f32x4 x = xr[tx][j];
                    const f32x4 ww = w[ty * TW_ + tx][j];
                    part += x[0] * ww[0] + x[1] * ww[1] + x[2] * ww[2] + x[3] * ww[3];
                }
                acc = fmaf(fr[tx], part, acc);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) {
            const size_t opix = out_pixel(p, b, oy, ox);
            float r = acc + bias;
            if (p.rowscale) r *= p.rowscale[opix];
            r = apply_act(r, p.act, p.slope);
            if (p.gate) r *= gate_factor(p, opix);
            if (p.accumulate) r += p.dst[opix];
            p.dst[opix] = r;
        }
    }
}
static bool to1w_ok(const IGemmParams& p) {
    static const bool off = getenv("TG_NO_TO1W") != nullptr;
    return !off && p.N == 1 && (p.C == 256 || p.C == 512) && ((p.TH == 4 && p.TW == 4) || (p.TH == 3 && p.TW == 3));
}

static bool to1_cfg_ok(int th, int tw) {
    return (th == 3 && tw == 3) || (th == 2 && tw == 2) || (th == 4 && tw == 4) || (th == 1 && tw == 1) || (th == 2 && tw == 1) ||
           (th == 1 && tw == 2);
}

bool smallconv_fwd_applies(const IGemmParams& p) {
    if (getenv("TG_NO_SMALLCONV")) return false;
    const int taps = p.TH * p.TW;
    if (p.C == 1 && p.N >= 64 && p.N % 64 == 0 && taps >= 1 && taps <= 64) return true;
    if (p.N == 1 && p.C == 64 && (p.OW % 4) == 0 && to1_cfg_ok(p.TH, p.TW)) return true;
    if (to1w_ok(p)) return true;
    return false;
}

static bool to1_fwd_lds_ok(const IGemmParams& p) {
    static const bool no_lds = getenv("TG_NO_TO1LDS") != nullptr;
    return !no_lds && p.TH == 3 && p.TW == 3 && p.ss == 1 && (p.tstep == 1 || p.tstep == -1) && p.OH >= T1_TH && p.OW >= T1_TW;
}
bool smallconv_bnin_fwd_ok(const IGemmParams& p) {
    return smallconv_fwd_applies(p) && p.N == 1 && p.C == 64 && !to1w_ok(p) && to1_fwd_lds_ok(p) && !p.amask;
}

bool smallconv_to1_multi_applies(const IGemmParams* cls, int ncls) {
    if (getenv("TG_NO_SMALLCONV") || getenv("TG_NO_TO1_MULTI") || ncls != 4) return false;
    for (int i = 0; i < ncls; ++i) {
        const IGemmParams& p = cls[i];
        if (p.N != 1 || p.C != 64 || p.TH != 2 || p.TW != 2 || (p.OW & 3) || p.M != cls[0].M || p.OH != cls[0].OH || p.OW != cls[0].OW) return false;
    }
    return true;
}
int smallconv_to1_multi_launch(const IGemmParams* cls, int ncls, hipStream_t s) {
    IGemmMulti pm = {};
    for (int i = 0; i < ncls; ++i) pm.c[i] = cls[i];
    {
        // one LDS patch for the four classes: they must read the same source tensor within a (T + 2)-pixel window, unmasked
        static const bool no_lds = getenv("TG_NO_TO1LDS") != nullptr || getenv("TG_NO_TO1_MULTI_LDS") != nullptr;
        int sy_min = 1 << 30, sx_min = 1 << 30, sy_max = -(1 << 30), sx_max = -(1 << 30);
        bool ok = !no_lds && cls[0].OH >= T1_TH && cls[0].OW >= T1_TW;
        for (int i = 0; i < ncls; ++i) {
            const IGemmParams& p = cls[i];
            ok = ok && p.src == cls[0].src && p.IH == cls[0].IH && p.IW == cls[0].IW && p.B == cls[0].B && p.ss == 1 && !p.amask &&
                 (p.tstep == 1 || p.tstep == -1);
            for (int t = 0; t < 2; ++t) {
                const int y = p.sy0 + t * p.tstep, x = p.sx0 + t * p.tstep;
                sy_min = y < sy_min ? y : sy_min; sy_max = y > sy_max ? y : sy_max;
                sx_min = x < sx_min ? x : sx_min; sx_max = x > sx_max ? x : sx_max;
            }
        }
        if (ok && sy_max - sy_min <= 2 && sx_max - sx_min <= 2) {
            const int tiles_x = cdiv(cls[0].OW, T1_TW), tiles_y = cdiv(cls[0].OH, T1_TH);
            const size_t lds = (size_t)T1_PH * T1_PW * 64 * sizeof(float);
            hipLaunchKernelGGL(to1conv64_multi22_lds_kernel, dim3(tiles_x * tiles_y * cls[0].B), dim3(256), lds, s, pm, tiles_x, tiles_y,
                               sy_min, sx_min);
            TG_CHECK_LAUNCH("to1conv64_multi22_lds_kernel");
            return TG_OK;
        }
    }
    int blocks = cdiv(cls[0].M / 4, 4 * 4);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(to1conv64_multi22_kernel, dim3(blocks, ncls), dim3(256), 0, s, pm);
    TG_CHECK_LAUNCH("to1conv64_multi22_kernel");
    return TG_OK;
}

#define TO1_CASE(TH_, TW_)                                                                           \
    if (p.TH == TH_ && p.TW == TW_) {                                                                \
        hipLaunchKernelGGL((to1conv64_kernel<TH_, TW_>), dim3(blocks), dim3(256), 0, s, p);          \
        TG_CHECK_LAUNCH("to1conv64_kernel");                                                         \
        return TG_OK;                                                                                \
    }

int smallconv_fwd_launch(const IGemmParams& p, hipStream_t s) {
    TG_REQUIRE(!p.in_bn.mean || smallconv_bnin_fwd_ok(p), "smallconv: BatchNorm-on-load is not available for this geometry");
    if (p.C == 1) {
        C1Geom q;
        q.tiles_x = cdiv(p.OW, C1_T);
        q.tiles_y = cdiv(p.OH, C1_T);
        const int sy_b = p.sy0 + (p.TH - 1) * p.tstep, sx_b = p.sx0 + (p.TW - 1) * p.tstep;
        q.sy_min = p.sy0 < sy_b ? p.sy0 : sy_b;
        q.sx_min = p.sx0 < sx_b ? p.sx0 : sx_b;
        q.PH = (C1_T - 1) * p.ss + (p.TH - 1) + 1;
        q.PW = (C1_T - 1) * p.ss + (p.TW - 1) + 1;
        dim3 grid(q.tiles_x * q.tiles_y * p.B, p.N / 64);
        static const bool no_mfma = getenv("TG_NO_C1MFMA") != nullptr;
        const bool al16 = ((reinterpret_cast<uintptr_t>(p.dst) | reinterpret_cast<uintptr_t>(p.bias) | reinterpret_cast<uintptr_t>(p.gate)) & 15) == 0;
        if (!no_mfma && al16 && p.TH == p.TW && (p.TH == 7 || p.TH == 4 || p.TH == 3)) {
            const int ks = (p.TH * p.TW + 1) / 2;
            const size_t lds = ((size_t)2 * ks * 68 + 4 * 32 * 36 + (size_t)q.PH * q.PW) * sizeof(float);
            if (p.TH == 7) hipLaunchKernelGGL((c1mfma_kernel<7, 7>), grid, dim3(256), lds, s, p, q);
            else if (p.TH == 4) hipLaunchKernelGGL((c1mfma_kernel<4, 4>), grid, dim3(256), lds, s, p, q);
            else hipLaunchKernelGGL((c1mfma_kernel<3, 3>), grid, dim3(256), lds, s, p, q);
            TG_CHECK_LAUNCH("c1mfma_kernel");
            return TG_OK;
        }
        const size_t lds = ((size_t)q.PH * q.PW + (size_t)p.TH * p.TW * 64) * sizeof(float);
        if (p.TH == 7 && p.TW == 7) hipLaunchKernelGGL((c1conv_kernel<7, 7>), grid, dim3(256), lds, s, p, q);
        else if (p.TH == 4 && p.TW == 4) hipLaunchKernelGGL((c1conv_kernel<4, 4>), grid, dim3(256), lds, s, p, q);
        else if (p.TH == 3 && p.TW == 3) hipLaunchKernelGGL((c1conv_kernel<3, 3>), grid, dim3(256), lds, s, p, q);
        else hipLaunchKernelGGL((c1conv_kernel<0, 0>), grid, dim3(256), lds, s, p, q);
        TG_CHECK_LAUNCH("c1conv_kernel");
        return TG_OK;
    }
    if (to1w_ok(p)) {
        int blocks = cdiv(p.M, 4 * 3);                 // ~3 outputs per wave: the register-resident weights are amortised
        if (blocks > 1024) blocks = 1024;
        if (blocks < 1) blocks = 1;
        if (p.TH == 4 && p.C == 512) hipLaunchKernelGGL((to1convw_kernel<4, 4, 2>), dim3(blocks), dim3(256), 0, s, p);
        else if (p.TH == 4) hipLaunchKernelGGL((to1convw_kernel<4, 4, 1>), dim3(blocks), dim3(256), 0, s, p);
        else if (p.C == 512) hipLaunchKernelGGL((to1convw_kernel<3, 3, 2>), dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((to1convw_kernel<3, 3, 1>), dim3(blocks), dim3(256), 0, s, p);
        TG_CHECK_LAUNCH("to1convw_kernel");
        return TG_OK;
    }
    if (to1_fwd_lds_ok(p)) {
        const int tiles_x = cdiv(p.OW, T1_TW), tiles_y = cdiv(p.OH, T1_TH);
        const int sy_b = p.sy0 + 2 * p.tstep, sx_b = p.sx0 + 2 * p.tstep;
        const int sy_min = p.sy0 < sy_b ? p.sy0 : sy_b, sx_min = p.sx0 < sx_b ? p.sx0 : sx_b;
        const size_t lds = (size_t)T1_PH * T1_PW * 64 * sizeof(float);
        if (p.in_bn.mean) hipLaunchKernelGGL(to1conv64_lds_kernel<true>, dim3(tiles_x * tiles_y * p.B), dim3(256), lds, s, p, tiles_x, tiles_y, sy_min, sx_min);
        else hipLaunchKernelGGL(to1conv64_lds_kernel<false>, dim3(tiles_x * tiles_y * p.B), dim3(256), lds, s, p, tiles_x, tiles_y, sy_min, sx_min);
        TG_CHECK_LAUNCH("to1conv64_lds_kernel");
        return TG_OK;
    }
    int blocks = cdiv(p.M / 4, 4 * 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    TO1_CASE(3, 3) TO1_CASE(2, 2) TO1_CASE(4, 4) TO1_CASE(1, 1) TO1_CASE(2, 1) TO1_CASE(1, 2)
    tg_set_error("smallconv: no to1conv configuration for taps %dx%d", p.TH, p.TW);
    return TG_ERR_ARG;
}

// ---- weight gradients ------------------------------------------------------------------------------------
// Cin == 1: dW[co][tap] = sum_pix dy[pix][co] * x[pix@tap]; lane = co.  Persistent workgroups walk 16x16 output
// tiles (patch staged in LDS per tile), accumulate K*K taps in registers and reduce across the 4 waves once.
template <int K>
__global__ __launch_bounds__(256) void c1wgrad_kernel(const WgradParams p, const C1Geom q, int ntiles, float* __restrict__ partial) {
    extern __shared__ float sm[];
    float* patch = sm;                             // [PH*PW]
    float* red = sm + ((q.PH * q.PW + 3) & ~3);    // [4][K][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + lane;
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.f;
    for (int tile0 = blockIdx.x; tile0 < ntiles; tile0 += gridDim.x) {
        int tile = tile0;
        const int txi = tile % q.tiles_x;
        tile /= q.tiles_x;
        const int tyi = tile % q.tiles_y, b = tile / q.tiles_y;
        const int oy0 = tyi * C1_T, ox0 = txi * C1_T;
        __syncthreads();
        c1_stage_patch(p.x, p.amask, patch, b, oy0 * p.stride - p.pad, ox0 * p.stride - p.pad, q.PH, q.PW, p.H, p.W);
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            const int ty_o = wave * 4 + r;
            const int oy = oy0 + ty_o;
            if (oy >= p.Ho) break;
#pragma unroll 4
            for (int tx_o = 0; tx_o < C1_T; ++tx_o) {
                const int ox = ox0 + tx_o;
                if (ox >= p.Wo) break;
                const float dyv = p.dy[(((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + n];
                const float* pb = patch + ty_o * p.stride * q.PW + tx_o * p.stride;
#pragma unroll
                for (int ky = 0; ky < K; ++ky)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) acc[ky * K + kx] = fmaf(dyv, pb[ky * q.PW + kx], acc[ky * K + kx]);
            }
        }
    }
    // cross-wave reduction, one kernel row at a time (keeps the LDS footprint small): partial[block][co][tap]
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        __syncthreads();
#pragma unroll
        for (int kx = 0; kx < K; ++kx) red[(wave * K + kx) * 64 + lane] = acc[ky * K + kx];
        __syncthreads();
        for (int i = threadIdx.x; i < K * 64; i += 256) {
            const int kx = i >> 6, c = i & 63;
            const float v = red[(0 * K + kx) * 64 + c] + red[(1 * K + kx) * 64 + c] + red[(2 * K + kx) * 64 + c] +
                            red[(3 * K + kx) * 64 + c];
            partial[((size_t)blockIdx.x * p.Cout + blockIdx.y * 64 + c) * (K * K) + ky * K + kx] = v;
        }
    }
}


// Cin == 1 weight gradient on the fp32 MFMA: dW[co][tap] = sum_pix dy[pix][co] * x[pix @ tap] is a GEMM with M = 64 output
// channels, N = taps (padded to 32s), K = pixels.  A[co][pix] is read straight from dy (a lane's 4 bytes are part of a
// coalesced 128-byte row of 32 channels), B[pix][tap] is the im2col view of the 1-channel LDS patch (one ds_read_b32 per lane:
// lane = tap, lane half = one of the step's two pixels).  The scalar-gather path of the generic wgrad kernel ran this at
// 0.7 TB/s (enc1: 113 us for 67 MB of dy; D conv0: 140 us for 134 MB).  Persistent workgroups over 16x16 output tiles,
// 4 waves x 64 pixels each; partial[block][co][tap], reduced by smallconv_slab_reduce in fixed order (deterministic).
// The tap tiles are padded to 32 columns and column NT is free for every kernel size here (9, 16, 49): its B operand is the
// constant 1, so the same MFMAs leave sum_pix dy[pix][co] -- the conv BIAS gradient -- in it (partial_db[block][co]; D conv0's
// bias gradient was a separate column-sum pass over the same 67 / 134 MB of dy).
template <int K>
__global__ __launch_bounds__(256) void c1wgrad_mfma_kernel(const WgradParams p, const C1Geom q, int ntiles, float* __restrict__ partial,
                                                           float* __restrict__ partial_db) {
    constexpr int NT = K * K, NTT = (NT + 31) / 32;
    static_assert(NT < NTT * 32, "no free column for the bias sum");
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* patch = sm;                                  // [PH*PW]
    float* red = sm + ((q.PH * q.PW + 3) & ~3);         // [64][NTT*32]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, li = lane & 31;
    const int co0 = blockIdx.y * 64;
    int tapoff[NTT];
    bool tapok[NTT];
    float tapfill[NTT];          // B value of a column without a tap: 1 in the bias column, 0 elsewhere
#pragma unroll
    for (int t = 0; t < NTT; ++t) {
        const int tap = 32 * t + li;
        tapok[t] = tap < NT;
        tapfill[t] = tap == NT ? 1.f : 0.f;
        const int tt = tapok[t] ? tap : 0;
        tapoff[t] = (tt / K) * q.PW + (tt % K);
    }
    f32x16 acc[2][NTT];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < NTT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;
    for (int tile0 = blockIdx.x; tile0 < ntiles; tile0 += gridDim.x) {
        int tile = tile0;
        const int txi = tile % q.tiles_x;
        tile /= q.tiles_x;
        const int tyi = tile % q.tiles_y, b = tile / q.tiles_y;
        const int oy0 = tyi * C1_T, ox0 = txi * C1_T;
        // this wave's 64 pixels: rows 4*wave .. 4*wave+3, 16 columns; K step s = pixels 2s, 2s+1 (lane half h), eight steps a batch.
        // The dy words of a batch are requested one batch ahead (the first one ahead of the patch staging): with the loads issued
        // and awaited inside the batch no memory request was in flight during its 32 MFMAs -- two waves per SIMD, nothing to hide
        // the round trip behind (enc1 37 us, D conv0 51 us for 13.6 us of matrix time)
        float an[8][2];
        auto load_a = [&](int s8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = 8 * s8 + u;
                const int ty_o = wave * 4 + (s >> 3), tx_o = ((2 * s) & 15) + h;
                const int oy = oy0 + ty_o, ox = ox0 + tx_o;
                const bool in = oy < p.Ho && ox < p.Wo;
                const float* dp = p.dy + (((size_t)b * p.Ho + (in ? oy : 0)) * p.Wo + (in ? ox : 0)) * p.Cout + co0 + li;
                // unconditional loads (the address is clamped into the image): a branch around each pair makes the outstanding-load
                // count unknowable to the compiler and every wait a vmcnt(0) -- which would also wait for the batch just requested
                const float v0 = dp[0], v1 = dp[32];
                an[u][0] = in ? v0 : 0.f;
                an[u][1] = in ? v1 : 0.f;
            }
        };
        load_a(0);
        __syncthreads();
        c1_stage_patch(p.x, p.amask, patch, b, oy0 * p.stride - p.pad, ox0 * p.stride - p.pad, q.PH, q.PW, p.H, p.W);
        __syncthreads();
#pragma unroll
        for (int s8 = 0; s8 < 4; ++s8) {
            float a[8][2], bv[8][NTT];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u][0] = an[u][0];
                a[u][1] = an[u][1];
            }
            if (s8 < 3) load_a(s8 + 1);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = 8 * s8 + u;
                const int ty_o = wave * 4 + (s >> 3), tx_o = ((2 * s) & 15) + h;
                const int pbase = ty_o * p.stride * q.PW + tx_o * p.stride;
#pragma unroll
                for (int t = 0; t < NTT; ++t) bv[u][t] = tapok[t] ? patch[pbase + tapoff[t]] : tapfill[t];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < NTT; ++t) acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][c], bv[u][t], acc[c][t], 0, 0, 0);
        }
    }
    // cross-wave reduction in a fixed order (wave 0 stores, waves 1..3 add), then partial[block][co][tap]
    constexpr int RP = NTT * 32 + 1;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < NTT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float* d = red + co * RP + 32 * t + li;
                        *d = w == 0 ? acc[c][t][r] : *d + acc[c][t][r];
                    }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * NT; i += 256) {
        const int co = i / NT, tap = i - co * NT;
        partial[((size_t)blockIdx.x * p.Cout + co0 + co) * NT + tap] = red[co * RP + tap];
    }
    if (partial_db && threadIdx.x < 64) partial_db[(size_t)blockIdx.x * p.Cout + co0 + threadIdx.x] = red[threadIdx.x * RP + NT];
}

// Cout == 1, C == 64: dW[tap][c] = sum_pix dy[pix] * x[pix@tap][c]; lane = (pixel of a 4-pixel row segment, channel quad)
template <int K>
__global__ __launch_bounds__(256) void to1wgrad64_kernel(const WgradParams p, float* __restrict__ partial, int quads_per_block) {
    __shared__ float red[4][K * K][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane >> 4, cq = lane & 15;
    f32x4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int quads = p.Mpix >> 2;
    const int q_begin = blockIdx.x * quads_per_block;
    const int q_end = min(quads, q_begin + quads_per_block);
    for (int q0 = q_begin + wave; q0 < q_end; q0 += 4) {
        const int q = __builtin_amdgcn_readfirstlane(q0);
        const int m0 = 4 * q;
        const int ox0 = m0 % p.Wo;
        const int t2 = m0 / p.Wo;
        const int oy = t2 % p.Ho, b = t2 / p.Ho;
        const int ox = ox0 + e;
        const float dyv = p.dy[m0 + e];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * p.stride - p.pad + ky;
            const bool vy = iy >= 0 && iy < p.H;
            const size_t rowbase = ((size_t)b * p.H + min(max(iy, 0), p.H - 1)) * p.W;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * p.stride - p.pad + kx;
                const bool vx = ix >= 0 && ix < p.W;
                const size_t pix = rowbase + min(max(ix, 0), p.W - 1);
                float f = (vy && vx) ? dyv : 0.f;
                if (p.amask) f *= p.amask[pix];
                const f32x4 x = *reinterpret_cast<const f32x4*>(p.x + pix * 64 + 4 * cq);
                acc[ky * K + kx] += f * x;
            }
        }
    }
    // fold the 4 pixel groups (lanes l, l^16, l^32), then the 4 waves through LDS: partial[block][tap][c]
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[t][j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[t][j] = v;
        }
        if (e == 0) *reinterpret_cast<f32x4*>(&red[wave][t][4 * cq]) = acc[t];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K * K * 64; i += 256) {
        const int t = i >> 6, c = i & 63;
        partial[((size_t)blockIdx.x * (K * K) + t) * 64 + c] = red[0][t][c] + red[1][t][c] + red[2][t][c] + red[3][t][c];
    }
}


// Cout == 1, C == 64, 3x3 stride 1: the weight gradient through the same LDS patch (see to1conv64_lds_kernel).  Persistent
// workgroups walk 4 x 32-pixel tiles and keep the 9 x 4-channel accumulators in registers; partial[block][tap][c].
template <bool BNIN>
__global__ __launch_bounds__(256) void to1wgrad64_lds_kernel(const WgradParams p, float* __restrict__ partial, int tiles_x, int tiles_y,
                                                             int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [T1_PH][T1_PW][64] + dy tile [T1_TH][T1_TW]
    float* dys = sm + T1_PH * T1_PW * 64;
    float* red = sm;                                                 // [4][9][64] after the tile loop
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = lane >> 4, cq = lane & 15;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NSLOT = T1_PH * T1_PW * 16, NIT = (NSLOT + 255) / 256;
    for (int tile0 = blockIdx.x; tile0 < ntiles; tile0 += gridDim.x) {
        int tile = tile0;
        const int txi = tile % tiles_x;
        tile /= tiles_x;
        const int tyi = tile % tiles_y, b = tile / tiles_y;
        const int oy0 = tyi * T1_TH, ox0 = txi * T1_TW;
        f32x4 v[NIT];
        float f[BNIN ? 1 : NIT];
        uint32_t inbits = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int slot = threadIdx.x + 256 * it, pp = slot >> 4;
            const int py = pp / T1_PW, px = pp - py * T1_PW;
            const int iy = oy0 - p.pad + py, ix = ox0 - p.pad + px;
            const bool in = slot < NSLOT && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const size_t pix = in ? ((size_t)b * p.H + iy) * p.W + ix : 0;
            v[it] = *reinterpret_cast<const f32x4*>(p.x + pix * 64 + 4 * (slot & 15));
            if constexpr (BNIN) inbits |= (in ? 1u : 0u) << it;
            else f[it] = !in ? 0.f : (p.amask ? p.amask[pix] : 1.f);
        }
        float dyv = 0.f;
        if (threadIdx.x < T1_TH * T1_TW) {
            const int oy = oy0 + threadIdx.x / T1_TW, ox = ox0 + threadIdx.x % T1_TW;
            if (oy < p.Ho && ox < p.Wo) dyv = p.dy[((size_t)b * p.Ho + oy) * p.Wo + ox];
        }
        if constexpr (BNIN) bn_in_stage<NIT>(p.in_bn, v, 4 * (threadIdx.x & 15));
        __syncthreads();                     // the previous tile's readers are done
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int slot = threadIdx.x + 256 * it;
            f32x4 o;
            if constexpr (BNIN) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (inbits >> it) & 1u ? v[it][k] : 0.f;
            } else {
                o = v[it] * f[it];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = f[it] != 0.f ? o[k] : 0.f;
            }
            if (slot < NSLOT) *reinterpret_cast<f32x4*>(sm + 4 * slot) = o;
        }
        if (threadIdx.x < T1_TH * T1_TW) dys[threadIdx.x] = dyv;
        __syncthreads();
        // lane (e, cq) takes FOUR consecutive pixels per segment of 16: the 3 x 6 patch pixels they share are read once (18 LDS
        // reads for 36 tap updates instead of one read per update: 99 -> 77 us on `final`, 3.6 TB/s).  The same regrouping in the
        // forward kernel changes nothing there (87 -> 90 us): it is not bound by its LDS reads; a persistent variant that
        // requests the next tile's patch ahead of the compute phase is slower (137 us: 26 more live registers per thread)
#pragma unroll 1
        for (int seg = 0; seg < T1_TW / 16; ++seg) {
            const int x0 = 16 * seg + 4 * e;
            const f32x4 d = *reinterpret_cast<const f32x4*>(dys + wave * T1_TW + x0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* pr = sm + (((wave + ky) * T1_PW) + x0) * 64 + 4 * cq;
                f32x4 X[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) X[c] = *reinterpret_cast<const f32x4*>(pr + c * 64);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ky * 3 + kx] += d[j] * X[j + kx];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float vv = acc[t][j];
            vv += __shfl_xor(vv, 16, 64);
            vv += __shfl_xor(vv, 32, 64);
            acc[t][j] = vv;
        }
        if (e == 0) *reinterpret_cast<f32x4*>(&red[(wave * 9 + t) * 64 + 4 * cq]) = acc[t];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * 64; i += 256) {
        const int t = i >> 6, c = i & 63;
        partial[((size_t)blockIdx.x * 9 + t) * 64 + c] = red[(0 * 9 + t) * 64 + c] + red[(1 * 9 + t) * 64 + c] + red[(2 * 9 + t) * 64 + c] +
                                                         red[(3 * 9 + t) * 64 + c];
    }
}

// second stage of the small wgrad kernels: ONE WAVE per output element -- lane l sums partial blocks l, l+64, ... (four
// independent loads per trip), then a fixed xor tree over the lanes: deterministic, and the ~1000 partials of an element
// are no longer a serial chain of dependent loads (that cost 74 us for 576 outputs)
__global__ __launch_bounds__(256) void smallconv_slab_reduce(const float* __restrict__ ws, float* __restrict__ out, size_t n,
                                                             int splits) {
    const size_t idx = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    double v = 0.0;
    if (idx < n) {
        int z = lane;
        for (; z + 192 < splits; z += 256) {
            const float a0 = ws[(size_t)z * n + idx], a1 = ws[(size_t)(z + 64) * n + idx];
            const float a2 = ws[(size_t)(z + 128) * n + idx], a3 = ws[(size_t)(z + 192) * n + idx];
            v += (double)a0 + (double)a1 + (double)a2 + (double)a3;
        }
        for (; z < splits; z += 64) v += (double)ws[(size_t)z * n + idx];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (idx < n && lane == 0) out[idx] = (float)v;
}


// Cout == 1, wide C (C % 256 == 0), stride 1: dW[tap][c] = sum_pix dy[pix] * x[pix@tap][c].  Workgroup = (image, 256-channel
// group, band of 4 output rows); wave = one output row, lane = channel quad.  Per kernel row ky the wave loads the input row
// segment of 8 outputs (8 + K - 1 pixels, all loads in flight together) ONCE into registers and feeds every (output, kx) pair
// from them; dy values are wave-uniform scalars.  partial[image * bands + band][tap][C].
template <int K>
__global__ __launch_bounds__(256) void to1wgradw_kernel(const WgradParams p, float* __restrict__ partial) {
    constexpr int OXC = 8, NX = OXC + K - 1;
    __shared__ float red[4][K * K][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, cg = blockIdx.y, oy = 4 * blockIdx.z + wave;
    f32x4 acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xb = p.x + (size_t)b * p.H * p.W * p.C + 256 * cg + 4 * lane;
    if (oy < p.Ho) {
        for (int ox0 = 0; ox0 < p.Wo; ox0 += OXC) {
            float dyv[OXC];
#pragma unroll
            for (int o = 0; o < OXC; ++o) dyv[o] = ox0 + o < p.Wo ? p.dy[((size_t)b * p.Ho + oy) * p.Wo + ox0 + o] : 0.f;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int iy = oy - p.pad + ky;
                if (iy < 0 || iy >= p.H) continue;                     // wave-uniform
                f32x4 xr[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const int ix = ox0 - p.pad + i;
                    const bool in = ix >= 0 && ix < p.W;
                    const size_t pix = (size_t)iy * p.W + (in ? ix : 0);
                    float f = in ? 1.f : 0.f;
                    if (p.amask) f *= p.amask[(size_t)b * p.H * p.W + pix];
                    xr[i] = f * *reinterpret_cast<const f32x4*>(xb + pix * p.C);
                }
#pragma unroll
                for (int o = 0; o < OXC; ++o)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) acc[ky * K + kx] += dyv[o] * xr[o + kx];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < K * K; ++t) *reinterpret_cast<f32x4*>(&red[wave][t][4 * lane]) = acc[t];
    __syncthreads();
    const size_t slab = (size_t)b * gridDim.z + blockIdx.z;
    for (int i = threadIdx.x; i < K * K * 256; i += 256) {
        const int t = i >> 8, c = i & 255;
        partial[(slab * (K * K) + t) * p.C + 256 * cg + c] = red[0][t][c] + red[1][t][c] + red[2][t][c] + red[3][t][c];
    }
}
static bool to1w_wgrad_ok(const WgradParams& p) {
    static const bool off = getenv("TG_NO_TO1W") != nullptr;
    return !off && p.Cout == 1 && (p.C % 256) == 0 && p.C <= 1024 && (p.k == 3 || p.k == 4) && p.stride == 1;
}

static int to1_wgrad_blocks(const WgradParams& p) {
    int blocks = cdiv(p.Mpix / 4, 64);
    if (blocks > 1024) blocks = 1024;    // 4 resident workgroups per CU: the loop is latency-bound with fewer
    if (blocks < 1) blocks = 1;
    return blocks;
}
static int c1_wgrad_blocks(const WgradParams& p) {
    int tiles = cdiv(p.Wo, C1_T) * cdiv(p.Ho, C1_T) * p.B;
    return tiles < 512 ? tiles : 512;
}

static bool to1_wgrad_lds_ok(const WgradParams& p) {
    static const bool off = getenv("TG_NO_TO1LDS") != nullptr;
    return !off && p.Cout == 1 && p.C == 64 && p.k == 3 && p.stride == 1 && p.pad == 1 && p.Ho >= T1_TH && p.Wo >= T1_TW;
}
bool smallconv_bnin_wgrad_ok(const WgradParams& p) { return !getenv("TG_NO_SMALLCONV") && (p.Wo % 4) == 0 && to1_wgrad_lds_ok(p) && !p.amask; }
bool smallconv_wgrad_applies(const WgradParams& p) {
    if (getenv("TG_NO_SMALLCONV")) return false;
    // Cin == 1 weight gradients measured faster on the MFMA wgrad kernel's scalar-gather path (0.11 vs 0.27 ms for
    // enc1 at 256^2/B=16); the dedicated kernel stays available behind TG_C1WGRAD=1 for experiments.
    if (p.C == 1 && p.Cout >= 64 && p.Cout % 64 == 0 && (p.k == 3 || p.k == 4 || p.k == 7) && !getenv("TG_NO_C1WGRAD_MFMA")) return true;
    if (p.Cout == 1 && p.C == 64 && (p.Wo % 4) == 0 && (p.k == 3 || p.k == 4)) return true;
    if (to1w_wgrad_ok(p)) return true;
    return false;
}
size_t smallconv_wgrad_ws_floats(const WgradParams& p) {
    if (to1w_wgrad_ok(p)) return (size_t)p.B * cdiv(p.Ho, 4) * p.k * p.k * p.C + 64;
    const int blocks = p.C == 1 ? c1_wgrad_blocks(p) : (to1_wgrad_lds_ok(p) ? 768 : to1_wgrad_blocks(p));
    return (size_t)blocks * p.Cout * p.k * p.k * p.C + (p.C == 1 ? (size_t)blocks * p.Cout : 0) + 64;     // (+ bias partials)
}
// db != nullptr: the launch may produce the bias gradient as well (*db_done = 1: the caller skips its column-sum pass)
int smallconv_wgrad_launch(const WgradParams& p, float* dw, float* ws, hipStream_t s, float* db, int* db_done) {
    if (db_done) *db_done = 0;
    float* pdb = nullptr;
    TG_REQUIRE(!p.in_bn.mean || smallconv_bnin_wgrad_ok(p), "smallconv: BatchNorm-on-load is not available for this geometry");
    int nb;
    if (p.C == 1) {
        C1Geom q;
        q.tiles_x = cdiv(p.Wo, C1_T);
        q.tiles_y = cdiv(p.Ho, C1_T);
        q.sy_min = q.sx_min = -p.pad;
        q.PH = q.PW = (C1_T - 1) * p.stride + p.k;
        const int ntiles = q.tiles_x * q.tiles_y * p.B;
        nb = c1_wgrad_blocks(p);
        dim3 grid(nb, p.Cout / 64);
        if (!getenv("TG_C1WGRAD")) {                 // default: the MFMA kernel
            const int ntt = (p.k * p.k + 31) / 32;
            const size_t lds = (((size_t)q.PH * q.PW + 3) / 4 * 4 + (size_t)64 * (ntt * 32 + 1)) * sizeof(float);
            static const bool no_db = getenv("TG_NO_C1WGRAD_BIAS") != nullptr;
            if (db && !no_db) pdb = ws + (size_t)nb * p.Cout * p.k * p.k * p.C;
            if (p.k == 7) hipLaunchKernelGGL((c1wgrad_mfma_kernel<7>), grid, dim3(256), lds, s, p, q, ntiles, ws, pdb);
            else if (p.k == 4) hipLaunchKernelGGL((c1wgrad_mfma_kernel<4>), grid, dim3(256), lds, s, p, q, ntiles, ws, pdb);
            else hipLaunchKernelGGL((c1wgrad_mfma_kernel<3>), grid, dim3(256), lds, s, p, q, ntiles, ws, pdb);
            TG_CHECK_LAUNCH("c1wgrad_mfma_kernel");
        } else {
        const size_t lds = (((size_t)q.PH * q.PW + 3) / 4 * 4 + (size_t)4 * p.k * 64) * sizeof(float);
        if (p.k == 7) hipLaunchKernelGGL((c1wgrad_kernel<7>), grid, dim3(256), lds, s, p, q, ntiles, ws);
        else if (p.k == 4) hipLaunchKernelGGL((c1wgrad_kernel<4>), grid, dim3(256), lds, s, p, q, ntiles, ws);
        else hipLaunchKernelGGL((c1wgrad_kernel<3>), grid, dim3(256), lds, s, p, q, ntiles, ws);
        TG_CHECK_LAUNCH("c1wgrad_kernel");
        }
    } else if (to1w_wgrad_ok(p)) {
        nb = p.B * cdiv(p.Ho, 4);
        if (p.k == 4) hipLaunchKernelGGL((to1wgradw_kernel<4>), dim3(p.B, p.C / 256, cdiv(p.Ho, 4)), dim3(256), 0, s, p, ws);
        else hipLaunchKernelGGL((to1wgradw_kernel<3>), dim3(p.B, p.C / 256, cdiv(p.Ho, 4)), dim3(256), 0, s, p, ws);
        TG_CHECK_LAUNCH("to1wgradw_kernel");
    } else if (to1_wgrad_lds_ok(p)) {
        const int tiles_x = cdiv(p.Wo, T1_TW), tiles_y = cdiv(p.Ho, T1_TH), ntiles = tiles_x * tiles_y * p.B;
        nb = ntiles < 768 ? ntiles : 768;           // three workgroups per CU (five would fit: 71 us against 65, more partials to reduce)
        const size_t lds = ((size_t)T1_PH * T1_PW * 64 + T1_TH * T1_TW) * sizeof(float);
        if (p.in_bn.mean) hipLaunchKernelGGL(to1wgrad64_lds_kernel<true>, dim3(nb), dim3(256), lds, s, p, ws, tiles_x, tiles_y, ntiles);
        else hipLaunchKernelGGL(to1wgrad64_lds_kernel<false>, dim3(nb), dim3(256), lds, s, p, ws, tiles_x, tiles_y, ntiles);
        TG_CHECK_LAUNCH("to1wgrad64_lds_kernel");
    } else {
        const int quads = p.Mpix / 4;
        const int blocks = to1_wgrad_blocks(p);
        const int qpb = cdiv(quads, blocks);
        nb = cdiv(quads, qpb);
        if (p.k == 4) hipLaunchKernelGGL((to1wgrad64_kernel<4>), dim3(nb), dim3(256), 0, s, p, ws, qpb);
        else hipLaunchKernelGGL((to1wgrad64_kernel<3>), dim3(nb), dim3(256), 0, s, p, ws, qpb);
        TG_CHECK_LAUNCH("to1wgrad64_kernel");
    }
    const size_t n = (size_t)p.Cout * p.k * p.k * p.C;
    hipLaunchKernelGGL(smallconv_slab_reduce, dim3((unsigned)cdiv64((int64_t)n, 4)), dim3(256), 0, s, ws, dw, n, nb);
    TG_CHECK_LAUNCH("smallconv_slab_reduce");
    if (pdb) {
        hipLaunchKernelGGL(smallconv_slab_reduce, dim3((unsigned)cdiv(p.Cout, 4)), dim3(256), 0, s, pdb, db, (size_t)p.Cout, nb);
        TG_CHECK_LAUNCH("smallconv_slab_reduce (bias)");
        if (db_done) *db_done = 1;
    }
    return TG_OK;
}
