// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 per SIMD as a function of waves/SIMD and accumulators per wave.
// hipcc -O3 -w --offload-arch=gfx950 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate && tools/micro/mfma_rate
// Result (MI355X): one wave per SIMD already issues at the full rate (148-156 TFLOP/s), with 1..16 accumulators, random
// operands and 141 KB of LDS allocated alike -- MFMA issue is never the limit of the Winograd kernels by itself.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 64 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, size_t lds) {
    float* out;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), lds, 0, out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), lds, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)grid * 4 * iters * 64;
    printf("NACC=%2d blocks/CU=%d  %.3f ms  %.1f TFLOP/s  (%.1f clk/MFMA/SIMD at 2.4 GHz)\n", NACC, blocks_per_cu, ms,
           mfma * 4096 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfma / (256.0 * 4)));
    hipFree(out);
}
template <int NACC>
__global__ __launch_bounds__(256) void k2(float* out, const float* in, int iters) {
    extern __shared__ float sm[];
    f32x16 acc[NACC];
    float a[NACC][4], b[NACC][4];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[i][e] = in[threadIdx.x + 256 * (i * 4 + e)]; b[i][e] = in[threadIdx.x + 256 * (i * 4 + e) + 7]; }
    }
    if (in[0] == 123.f) sm[threadIdx.x] = 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[i][e], acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run2(int blocks_per_cu, size_t lds, int randomize) {
    float *out, *in;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    hipMalloc(&in, 65536 * sizeof(float));
    float* h = (float*)malloc(65536 * 4);
    for (int i = 0; i < 65536; ++i) h[i] = randomize ? (float)rand() / RAND_MAX - 0.5f : 1.0f;
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k2<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int iters = 2000 * 16 / NACC, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k2<NACC>, dim3(grid), dim3(256), lds, 0, out, in, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k2<NACC>, dim3(grid), dim3(256), lds, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)grid * 4 * iters * 4 * NACC;
    printf("k2 NACC=%2d blocks/CU=%d lds=%zu rand=%d  %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, lds, randomize, ms, mfma * 4096 / ms / 1e9);
    hipFree(out); hipFree(in); free(h);
}
int main() {
    run2<16>(1, 0, 0); run2<16>(1, 0, 1); run2<16>(1, 141440, 1); run2<8>(2, 59000, 1); run2<8>(2, 0, 1); run2<4>(2, 0, 1); run2<4>(2, 0, 0);
    run<4>(1, 0); run<8>(1, 0); run<16>(1, 0);
    run<4>(2, 0); run<8>(2, 0);
    run<4>(4, 0);
    run<1>(1, 0); run<2>(1, 0); run<1>(4, 0);
    return 0;
}
