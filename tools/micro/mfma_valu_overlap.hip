// Micro-benchmark: do v_mfma_f32_32x32x2_f32 and plain fp32 VALU instructions overlap on a gfx950 SIMD?
// hipcc -O3 -w --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap && tools/micro/mfma_valu_overlap
// Each wave runs 64 MFMAs per iteration with NV independent v_fma_f32 placed (by sched_barrier) between consecutive MFMAs;
// 1 or 2 waves per SIMD.  If the two kinds of instruction overlap, time stays flat until NV * 4 cycles exceeds the 64-cycle
// MFMA; if the fp32 MFMA executes on the vector ALU's own FMA lanes, time grows by ~NV * issue cycles per MFMA from NV = 1.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    f32x16 acc[4];
    float v[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x + 64 * i];
    const float a = in[threadIdx.x], b = in[threadIdx.x + 1], c = in[5], d = in[6];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[(m * NV + j) & 15] = __builtin_fmaf(v[(m * NV + j) & 15], c, d);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV>
void run(int blocks_per_cu) {
    float *out, *in;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    hipMalloc(&in, 65536 * sizeof(float));
    hipMemset(in, 0, 65536 * 4);
    const int iters = 1000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, out, in, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)blocks_per_cu * iters * 64;
    printf("NV=%2d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s   %.1f ns per MFMA per SIMD\n", NV, blocks_per_cu, ms,
           (double)grid * 4 * iters * 64 * 4096 / ms / 1e9, ms * 1e6 / mfma_per_simd);
    hipFree(out); hipFree(in);
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>(w); run<1>(w); run<2>(w); run<4>(w); run<8>(w); run<12>(w); run<16>(w);
    }
    return 0;
}
