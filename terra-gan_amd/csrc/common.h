// Shared host/device helpers for libterragan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "terragan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

void tg_set_error(const char* fmt, ...);

#define TG_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            tg_set_error(__VA_ARGS__);   \
            return TG_ERR_ARG;           \
        }                                \
    } while (0)

#define TG_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            tg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return TG_ERR_LAUNCH;                                                    \
        }                                                                            \
    } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Memory-bound kernels: cap the grid and grid-stride the rest (256 CUs x 8 blocks).
static inline int ew_grid(int64_t work_items, int block) {
    int64_t g = cdiv64(work_items, block);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one rounding sequence for BatchNorm's affine map wherever it is applied (bn_act_fwd kernels, BN-on-load staging): sub, mul, fma
__device__ __forceinline__ float bn_affine(float v, float m, float r, float g, float b) {
    return __fmaf_rn(__fmul_rn(__fsub_rn(v, m), r), g, b);
}
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == TG_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == TG_ACT_LEAKY) return v > 0.f ? v : v * slope;
    return v;
}

// Epilogue helper.  The MFMA C layout gives a lane ONE column and 16 rows of a 32x32 tile, i.e. 16 single-dword
// stores per tile and lane; such an epilogue is store-ISSUE bound (it cost the patch kernel ~6 K-steps per workgroup).
// The wave bounces the tile through its own 32x36-float LDS scratch and comes back with 4 consecutive columns of one
// row per lane: emit(row_in_tile, col_in_tile, f32x4) is called 4 times per lane -> 4 x 16-byte stores instead of
// 16 x 4-byte ones.  A wave reads back only what it wrote itself (LDS ops of one wave complete in order): no barrier.
// as tile_rows4, emit(t, row_in_tile, col_in_tile, f32x4) with the call index t = 0..3 (a constant after unrolling)
template <class F>
__device__ __forceinline__ void tile_rows4i(float* scratch, const f32x16& acc, int lane, F&& emit) {
#pragma unroll
    for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[r];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rr = (lane >> 3) + 8 * t;
        emit(t, rr, 4 * (lane & 7), *reinterpret_cast<const f32x4*>(scratch + rr * 36 + 4 * (lane & 7)));
    }
}
template <class F>
__device__ __forceinline__ void tile_rows4(float* scratch, const f32x16& acc, int lane, F&& emit) {
#pragma unroll
    for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[r];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int rr = (lane >> 3) + 8 * t;
        emit(rr, 4 * (lane & 7), *reinterpret_cast<const f32x4*>(scratch + rr * 36 + 4 * (lane & 7)));
    }
}
