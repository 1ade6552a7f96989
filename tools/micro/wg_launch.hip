// Micro-benchmark: cost of dispatching workgroups shaped like wino_kernel's (512 threads, 141 KB dynamic LDS, 256 VGPRs):
// time of a grid of N empty workgroups / (N / 256) = dispatch + teardown time per workgroup slot on a CU.
// hipcc -O3 -w --offload-arch=gfx950 tools/micro/wg_launch.hip -o tools/micro/wg_launch && tools/micro/wg_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void k(float* out, int spin) {
    extern __shared__ float sm[];
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < spin; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.f, 1.f, acc[i], 0, 0, 0);
    sm[threadIdx.x] = acc[0][0];
    __syncthreads();
    float s = sm[(threadIdx.x + 1) & 511];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i][3];
    if (s == 12345.f) out[blockIdx.x] = s;
}
int main() {
    float* out;
    hipMalloc(&out, 1 << 20);
    const size_t lds = 141440;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int spin : {0, 32, 128}) {          // spin = 8-MFMA rounds per wave: 32 rounds = one 8-step K loop (256 MFMAs / wave)
        for (int grid : {256, 8192}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, out, spin);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, 0, out, spin);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("spin=%3d grid=%5d  %.1f us total, %.2f us per workgroup round\n", spin, grid, ms * 1e3, ms * 1e3 / (grid / 256.0));
        }
    }
    return 0;
}
