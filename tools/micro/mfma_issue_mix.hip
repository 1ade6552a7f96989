// Micro-benchmark: what does ONE instruction of another kind, issued between consecutive v_mfma_f32_32x32x2_f32 of the same
// wave, cost the matrix pipe on gfx950?  (Follow-up to mfma_valu_overlap.hip, which showed that v_fma_f32 does not overlap.)
// hipcc -O3 -w --offload-arch=gfx950 tools/micro/mfma_issue_mix.hip -o tools/micro/mfma_issue_mix && tools/micro/mfma_issue_mix
// KIND: 0 v_fma_f32   1 v_pk_fma_f32 (two fp32 FMAs per lane)   2 v_add_u32   3 v_cndmask_b32   4 ds_read_b128
//       5 ds_write_b128   6 s_add_u32 (scalar)   7 v_mov_b32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    __shared__ f32x4 lds[512];
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[8];
    f32x2 p[8];
    f32x4 w[4];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = in[threadIdx.x + 64 * i]; p[i] = f32x2{v[i], v[i]}; u[i] = threadIdx.x + i; }
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = f32x4{v[i], v[i], v[i], v[i]};
    lds[threadIdx.x] = w[0]; lds[threadIdx.x + 256] = w[1];
    __syncthreads();
    const float a = in[threadIdx.x], b = in[threadIdx.x + 1], c = in[5], d = in[6];
    const f32x2 c2{c, c}, d2{d, d};
    unsigned sacc = iters;
    const f32x4* lp = lds + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 64; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int s = (m * NV + j) & 7;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[s]) : "v"(c), "v"(d));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[s]) : "v"(c2), "v"(d2));
                if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[s]) : "v"(u[(s + 1) & 7]));
                if (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[s]) : "v"(u[(s + 1) & 7]) : );
                if (KIND == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(w[s & 3]) : "v"((unsigned)(threadIdx.x * 16)) : "memory");
                if (KIND == 5) asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)(threadIdx.x * 16)), "v"(w[s & 3]) : "memory");
                if (KIND == 6) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(u[s]) : "v"(u[(s + 1) & 7]));
            }
            if (KIND == 4 || KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = (float)sacc;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1] + (float)u[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += w[i][0] + w[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int NV>
double run(int blocks_per_cu) {
    float *out, *in;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    hipMalloc(&in, 65536 * sizeof(float));
    hipMemset(in, 0, 65536 * 4);
    const int iters = 500, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(grid), dim3(256), 0, 0, out, in, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NV>), dim3(grid), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out); hipFree(in);
    return ms * 1e6 / ((double)blocks_per_cu * iters * 64);       // ns per MFMA per SIMD
}
template <int KIND>
void kind(const char* name) {
    for (int w = 1; w <= 2; ++w) {
        const double t0 = run<KIND, 0>(w), t1 = run<KIND, 1>(w), t2 = run<KIND, 2>(w), t4 = run<KIND, 4>(w);
        printf("%-14s waves/SIMD=%d  ns per MFMA: NV=0 %.2f  1 %.2f  2 %.2f  4 %.2f   -> %.2f ns per extra instruction\n", name, w, t0, t1,
               t2, t4, (t4 - t0) / 4);
    }
}
int main() {
    kind<0>("v_fma_f32");
    kind<1>("v_pk_fma_f32");
    kind<2>("v_add_u32");
    kind<3>("v_cndmask_b32");
    kind<7>("v_mov_b32");
    kind<4>("ds_read_b128");
    kind<5>("ds_write_b128");
    kind<6>("s_add_u32");
    return 0;
}
