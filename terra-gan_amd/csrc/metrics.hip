// Logging-side kernels of the "next" rows: validation / batch quality metrics and the input pipeline.
//
//   tg_quality_metrics   one pass over (pred, target, mask) -> the scalars the reference logs every log_interval:
//                        calculate_boundary_quality (mvp_gan/src/evaluation/metrics.py:79-133: boundary MSE / PSNR over the
//                        3x3 morphological band, mean-|difference| "gradient" proxy) and the tracker's PSNR, 11x11 avg-pool
//                        SSIM, L1 and L2 (utils/experiment_tracking.py:176-231, = MaskEvaluator._calculate_psnr/_ssim,
//                        evaluation/metrics.py:47-76).  The reference runs ~25 ATen ops and 5 host syncs for these.
//   tg_u8_to_tiles       uint8 tile shards -> fp32 image (/255) and binarised mask (>0), the arithmetic of
//                        mvp_gan/src/utils/dataset.py:35-37 after the resize, on the device: 1 byte per pixel crosses PCIe
//                        instead of 4.
#include <math.h>

#include "common.h"

static inline hipStream_t S(tg_stream_t s) { return (hipStream_t)s; }

constexpr int QT = 32;              // output tile
constexpr int QR = 5;               // 11x11 window radius
constexpr int QP = QT + 2 * QR;     // 42: tile + halo
constexpr int QPP = QP + 1;         // LDS row pitch of the patches
constexpr int QN = 9;               // partial sums per block

// 3x3 morphological band of the mask, max_pool2d semantics (out-of-image taps ignored): metrics.py:88-90
__device__ __forceinline__ float q_band(const float* __restrict__ m, int64_t img, int y, int x, int H, int W) {
    float mx = -INFINITY, mn = INFINITY;
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const float v = m[(img * H + yy) * W + xx];
            mx = fmaxf(mx, v);
            mn = fminf(mn, v);
        }
    }
    const float d = mx - (1.f - (1.f - mn));
    return fminf(fmaxf(d, 0.f), 1.f);
}

// One 256-thread workgroup per 32x32 tile of one image.  The (tile + 5-pixel halo) patches of pred and target are staged in
// LDS with ZERO padding outside the image -- avg_pool2d(padding=5) pads with zeros and divides by 121 everywhere
// (count_include_pad) -- then the five 11x11 window sums (p, t, pp, tt, pt) are formed separably: a horizontal pass into
// LDS, a vertical pass in registers.  The other sums ride along on the interior pixels.
__global__ __launch_bounds__(256) void quality_tile_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                           const float* __restrict__ m, int H, int W, int tiles_x, int tiles_y,
                                                           double* __restrict__ partial) {
    __shared__ float Ps[QP][QPP], Ts[QP][QPP];
    __shared__ float Hs[5][QP][QT + 1];
    __shared__ double red[QN][4];
    const int tid = threadIdx.x;
    int tile = blockIdx.x;
    const int tx = tile % tiles_x;
    tile /= tiles_x;
    const int ty = tile % tiles_y;
    const int64_t img = tile / tiles_y;
    const int y0 = ty * QT, x0 = tx * QT;
    const float* pi = p + img * H * W;
    const float* ti = t + img * H * W;
    for (int i = tid; i < QP * QP; i += 256) {
        const int r = i / QP, c = i - r * QP;
        const int y = y0 + r - QR, x = x0 + c - QR;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        Ps[r][c] = in ? pi[(int64_t)y * W + x] : 0.f;
        Ts[r][c] = in ? ti[(int64_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < QP * QT; i += 256) {          // horizontal 11-sums for all 42 rows
        const int r = i / QT, c = i - r * QT;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * QR + 1; ++k) {
            const float a = Ps[r][c + k], b = Ts[r][c + k];
            s0 += a; s1 += b; s2 += a * a; s3 += b * b; s4 += a * b;
        }
        Hs[0][r][c] = s0; Hs[1][r][c] = s1; Hs[2][r][c] = s2; Hs[3][r][c] = s3; Hs[4][r][c] = s4;
    }
    __syncthreads();
    double q[QN] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f, cnt = 121.f;     // avg_pool2d divides the window sum by 121
    for (int i = tid; i < QT * QT; i += 256) {
        const int r = i / QT, c = i - r * QT;
        const int y = y0 + r, x = x0 + c;
        if (y >= H || x >= W) continue;
        float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 2 * QR + 1; ++k)
#pragma unroll
            for (int j = 0; j < 5; ++j) s[j] += Hs[j][r + k][c];
        const float mu1 = s[0] / cnt, mu2 = s[1] / cnt;
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float sg1 = s[2] / cnt - mu1_sq, sg2 = s[3] / cnt - mu2_sq, sg12 = s[4] / cnt - mu12;
        q[0] += (double)(((2.f * mu12 + C1) * (2.f * sg12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sg1 + sg2 + C2)));
        const float pv = Ps[r + QR][c + QR], tv = Ts[r + QR][c + QR];
        const float d = pv - tv;
        q[1] += (double)(d * d);
        q[2] += (double)fabsf(d);
        const float bd = q_band(m, img, y, x, H, W);
        const float db = d * bd;
        q[3] += (double)(db * db);
        q[4] += (double)bd;
        if (y + 1 < H) {
            q[5] += (double)fabsf(Ps[r + QR + 1][c + QR] - pv);
            q[7] += (double)fabsf(Ts[r + QR + 1][c + QR] - tv);
        }
        if (x + 1 < W) {
            q[6] += (double)fabsf(Ps[r + QR][c + QR + 1] - pv);
            q[8] += (double)fabsf(Ts[r + QR][c + QR + 1] - tv);
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < QN; ++i) {
        const double v = wave_sum_d(q[i]);
        if (lane == 0) red[i][wave] = v;
    }
    __syncthreads();
    if (tid < QN) partial[(size_t)blockIdx.x * QN + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
}

// out[0..8] = mse, psnr, ssim, l1, l2, boundary_mse, boundary_psnr, boundary_gradient_diff, sum(band)
__global__ void quality_final_kernel(const double* __restrict__ partial, int nblocks, int64_t imgs, int H, int W,
                                     float* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;       // one wave, fixed order: deterministic
    double s[QN];
#pragma unroll
    for (int i = 0; i < QN; ++i) s[i] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64)
#pragma unroll
        for (int i = 0; i < QN; ++i) s[i] += partial[(size_t)b * QN + i];
#pragma unroll
    for (int i = 0; i < QN; ++i) s[i] = wave_sum_d(s[i]);
    if (threadIdx.x != 0) return;
    const double n = (double)imgs * H * W;
    const float mse = (float)(s[1] / n);
    out[0] = mse;
    out[1] = mse == 0.f ? INFINITY : 20.f * log10f(1.f / sqrtf(mse));          // experiment_tracking.py:199-204
    out[2] = (float)(s[0] / n);
    out[3] = (float)(s[2] / n);
    out[4] = sqrtf(mse);
    const float band = (float)s[4];
    out[8] = band;
    if (band < 1e-6f) {                                                          // metrics.py:93-98
        out[5] = 0.f; out[6] = 0.f; out[7] = 0.f;
        return;
    }
    const float bmse = (float)(s[3] / n);                                        // mean over ALL elements (metrics.py:101)
    out[5] = bmse;
    out[6] = 10.f * log10f(1.f / (bmse + 1e-6f));
    const double nh = (double)imgs * (H - 1) * W, nw = (double)imgs * H * (W - 1);
    const float pd = (float)(s[5] / nh) + (float)(s[6] / nw), td = (float)(s[7] / nh) + (float)(s[8] / nw);
    out[7] = fabsf(pd - td);
}

static int q_tiles(int64_t imgs, int H, int W) { return (int)(imgs * cdiv(H, QT) * cdiv(W, QT)); }
extern "C" size_t tg_quality_metrics_ws_bytes(int64_t imgs, int H, int W) {
    return (size_t)q_tiles(imgs, H, W) * QN * sizeof(double) + 64;
}
extern "C" int tg_quality_metrics(const float* pred, const float* target, const float* mask, int64_t imgs, int H, int W,
                                  float* out9, float* ws, size_t ws_bytes, tg_stream_t stream) {
    TG_REQUIRE(pred && target && mask && out9 && ws, "tg_quality_metrics: null pointer");
    TG_REQUIRE(imgs > 0 && H > 1 && W > 1, "tg_quality_metrics: bad dims");
    TG_REQUIRE(imgs * cdiv(H, QT) * cdiv(W, QT) < (1ll << 30), "tg_quality_metrics: too many tiles");
    TG_REQUIRE(ws_bytes >= tg_quality_metrics_ws_bytes(imgs, H, W), "tg_quality_metrics: workspace too small");
    TG_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 7) == 0, "tg_quality_metrics: workspace must be 8-byte aligned");
    const int grid = q_tiles(imgs, H, W);
    double* partial = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(quality_tile_kernel, dim3(grid), dim3(256), 0, S(stream), pred, target, mask, H, W, cdiv(W, QT), cdiv(H, QT),
                       partial);
    TG_CHECK_LAUNCH("quality_tile_kernel");
    hipLaunchKernelGGL(quality_final_kernel, dim3(1), dim3(64), 0, S(stream), partial, grid, imgs, H, W, out9);
    TG_CHECK_LAUNCH("quality_final_kernel");
    return TG_OK;
}

// ---- uint8 tile shards -> fp32 tiles -----------------------------------------------------------------------------
// image = u8 / 255 (IEEE fp32 division, exactly numpy's `a.astype(float32) / 255.0`), mask = u8 > 0 (dataset.py:35-37).
// 16 pixels per thread: one 16-byte load, four 16-byte stores per output.
__global__ __launch_bounds__(256) void u8_to_tiles_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ msk,
                                                          int64_t n, float* __restrict__ out_img, float* __restrict__ out_msk) {
    const int64_t n16 = n >> 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        if (img) {
            const uint4 v = reinterpret_cast<const uint4*>(img)[i];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (float)((w[k] >> (8 * e)) & 0xffu) / 255.0f;
                reinterpret_cast<f32x4*>(out_img)[4 * i + k] = o;
            }
        }
        if (msk) {
            const uint4 v = reinterpret_cast<const uint4*>(msk)[i];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ((w[k] >> (8 * e)) & 0xffu) ? 1.f : 0.f;
                reinterpret_cast<f32x4*>(out_msk)[4 * i + k] = o;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 15)) {       // ragged tail
        const int64_t i = (n16 << 4) + threadIdx.x;
        if (img) out_img[i] = (float)img[i] / 255.0f;
        if (msk) out_msk[i] = msk[i] ? 1.f : 0.f;
    }
}
extern "C" int tg_u8_to_tiles(const uint8_t* img_u8, const uint8_t* mask_u8, int64_t n, float* img_f32, float* mask_f32,
                              tg_stream_t stream) {
    TG_REQUIRE(n > 0 && (img_u8 || mask_u8), "tg_u8_to_tiles: nothing to do");
    TG_REQUIRE((!img_u8 || img_f32) && (!mask_u8 || mask_f32), "tg_u8_to_tiles: missing output");
    TG_REQUIRE(((reinterpret_cast<uintptr_t>(img_u8) | reinterpret_cast<uintptr_t>(mask_u8) | reinterpret_cast<uintptr_t>(img_f32) |
                 reinterpret_cast<uintptr_t>(mask_f32)) & 15) == 0, "tg_u8_to_tiles: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(u8_to_tiles_kernel, dim3(ew_grid(cdiv64(n, 16), 256)), dim3(256), 0, S(stream), img_u8, mask_u8, n, img_f32,
                       mask_f32);
    TG_CHECK_LAUNCH("u8_to_tiles_kernel");
    return TG_OK;
}
