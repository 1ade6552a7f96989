// Micro-benchmark: the MFMA skeleton of wino_kernel's K loop and nothing else -- 512-thread workgroups (2 waves per SIMD), per
// wave 8 accumulator tiles x 4 v_mfma_f32_32x32x2_f32 per step, operands (a) constant registers, (b) fetched per transform
// point from LDS with ds_read_b128 one point ahead, (c) the same plus one / two workgroup barriers per step.
// hipcc -O3 -w --offload-arch=gfx950 tools/micro/wino_loop_model.hip -o tools/micro/wino_loop_model && tools/micro/wino_loop_model
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <class F, int... T>
__device__ __forceinline__ void static_for(std::integer_sequence<int, T...>, F&& f) { (f(std::integral_constant<int, T>{}), ...); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MODE 0: register operands   1: LDS fragments   2: + 1 barrier/step   3: + 2 barriers/step
//      4: + the 8 transform reads (ds_read_b128)   5: + 32 transform VALU   6: + the 10 staging writes (ds_write_b128)
//      7: + the 6 global loads (16 B per lane) of the step after next, feeding the writes
//      8: as 7 with 6 writes only (the U image's 4 writes and their loads dropped)
//      9: as 8 + the U image fetched by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), waited for before the barrier
//     10: mode 5 + 32 ds_write_b32 per wave (the transposing image writes of the wgrad kernels)
//     11: mode 5 + 8 ds_write_b128 per wave (the same bytes)      12: mode 11 + 36 ds_read_b32 per wave
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int steps, const float* big, size_t big_floats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mw = wave & 1, nw = (wave >> 1) & 1, xh = wave >> 2;
    for (int i = tid; i < 32768; i += 512) smem[i] = in[i & 1023];
    __syncthreads();
    const int arow = 32 * mw + (lane & 31), brow = 32 * nw + (lane & 31);
    const int a_off = (8 * xh) * 512 + arow * 8 + ((((lane >> 5) ^ (arow >> 4)) & 1) << 2);
    const int b_off = 16384 + (8 * xh) * 512 + brow * 8 + ((((lane >> 5) ^ (brow >> 4)) & 1) << 2);
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 fa[2], fb[2];
    fa[0] = *reinterpret_cast<const f32x4*>(smem + a_off);
    fb[0] = *reinterpret_cast<const f32x4*>(smem + b_off);
    fa[1] = fa[0]; fb[1] = fb[0];
    f32x4 td[8], tt[4], rg[6];
#pragma unroll
    for (int i = 0; i < 8; ++i) td[i] = fa[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) tt[i] = fb[0];
#pragma unroll
    for (int i = 0; i < 6; ++i) rg[i] = fa[0];
    const float s1 = in[7];
    const float* tsrc = smem + 28672 + (lane >> 1) * 16 + 4 * (lane & 1) + wave * 8;
    float* wdst = smem + 24576 + tid * 4;           // scratch area of the writes (not read by the MFMA fragments)
    const float* gp = big + (size_t)blockIdx.x * 65536 + tid * 4;
    for (int c = 0; c < steps; ++c) {
        const float* Vc = smem + a_off;
        const float* Uc = smem + b_off;
        const float* gc = gp + (size_t)(c & 63) * 16777216 % big_floats;
        static_for(std::make_integer_sequence<int, 32>{}, [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int xi = t >> 2, e = t & 3;
            if constexpr (MODE >= 3 && t == 4) lds_barrier();
            if constexpr (MODE >= 2 && t == 26) lds_barrier();
            if constexpr (MODE >= 6 && MODE <= 9 && t < 2) *reinterpret_cast<f32x4*>(wdst + 2048 * t) = rg[t];
            if constexpr (MODE >= 7 && MODE <= 9 && t >= 2 && t < 4) rg[t - 2] = *reinterpret_cast<const f32x4*>(gc + 2048 * (t - 2));
            if constexpr (MODE >= 4 && t >= 5 && t < 9) {
                td[t - 5] = *reinterpret_cast<const f32x4*>(tsrc + (t - 5) * 8);
                td[4 + t - 5] = *reinterpret_cast<const f32x4*>(tsrc + 512 + (t - 5) * 8);
            }
            if constexpr (MODE >= 5 && t >= 8 && t < 12) tt[t - 8] = td[t - 8] + td[4 + t - 8] * s1;
            if constexpr (MODE >= 5 && t >= 14 && t < 18) {
                constexpr int kk = t - 14;
                td[kk] = kk == 0 ? tt[0] - tt[2] : kk == 1 ? tt[1] + tt[2] : kk == 2 ? tt[2] - tt[1] : tt[1] - tt[3];
            }
            if constexpr (MODE >= 6 && MODE <= 9 && t >= 14 && t < 18) *reinterpret_cast<f32x4*>(wdst + 512 * (t - 14)) = td[t - 14];
            if constexpr (MODE >= 6 && MODE <= 7 && t >= 18 && t < 22) *reinterpret_cast<f32x4*>(wdst + 512 * (t - 18) + 64) = rg[2 + t - 18];
            if constexpr (MODE == 7 && t >= 22 && t < 26) rg[2 + t - 22] = *reinterpret_cast<const f32x4*>(gc + 4096 + 2048 * (t - 22));
            if constexpr (MODE == 9 && t == 25) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // the DMA of the previous step has landed
            if constexpr (MODE == 9 && t >= 27 && t < 31)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gc + 4096 + 2048 * (t - 27)),
                                                 (__attribute__((address_space(3))) void*)(smem + 24576 + 2048 + (wave * 64 + 512 * (t - 27)) * 4), 16, 0, 0);
            if constexpr (MODE == 10) *reinterpret_cast<float*>(wdst + 2048 * (t & 3) + (t >> 2) * 4 + (lane & 3)) = td[t & 7][t & 3];
            if constexpr ((MODE == 11 || MODE == 12) && (t & 3) == 1) *reinterpret_cast<f32x4*>(wdst + 512 * (t >> 2)) = td[t >> 2];
            if constexpr (MODE == 12) {
                td[t & 7][t & 3] += tsrc[1024 + t * 64 + lane];
                if constexpr (t < 4) td[t][3 - t] += tsrc[3072 + t * 64 + lane];
            }
            if constexpr (MODE >= 1 && e == 0) {
                fa[(xi + 1) & 1] = *reinterpret_cast<const f32x4*>(Vc + ((xi + 1) & 7) * 512);
                fb[(xi + 1) & 1] = *reinterpret_cast<const f32x4*>(Uc + ((xi + 1) & 7) * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[xi & 1][e], fb[xi & 1][e], acc[xi], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += td[i][0] + td[i][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += tt[i][1];
#pragma unroll
    for (int i = 0; i < 6; ++i) s += rg[i][2];
    out[blockIdx.x * 512 + tid] = s;
}
template <int MODE>
void run(const char* name) {
    float *out, *in, *big;
    const size_t big_floats = (size_t)1 << 28;          // 1 GB: the global loads stream, they do not hit in L2
    hipMalloc(&big, big_floats * sizeof(float));
    hipMemset(big, 0, big_floats * 4);
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&in, 65536 * sizeof(float));
    hipMemset(in, 0, 65536 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int steps = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 131072, 0, out, in, 10, big, big_floats);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 131072, 0, out, in, steps, big, big_floats);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_step = ms * 1e6 / steps;
    printf("%-44s %.1f ns per step (64 MFMAs per SIMD)  %.2f ns per MFMA  %.1f TFLOP/s\n", name, ns_step, ns_step / 64,
           256.0 * 8 * 32 * steps * 4096 / ms / 1e9);
    hipFree(out); hipFree(in); hipFree(big);
}
int main() {
    run<0>("register operands");
    run<1>("LDS fragments, one point ahead");
    run<2>("LDS fragments + 1 barrier per step");
    run<3>("LDS fragments + 2 barriers per step");
    run<4>("+ 8 transform ds_read_b128 per wave");
    run<5>("+ 32 transform VALU per wave");
    run<6>("+ 10 staging ds_write_b128 per wave");
    run<7>("+ 6 global_load_dwordx4 per wave");
    run<8>("6 writes + 2 loads (U path dropped)");
    run<9>("6 writes + 2 loads + U by LDS-DMA");
    run<10>("VALU + 32 ds_write_b32 per wave");
    run<11>("VALU + 8 ds_write_b128 per wave");
    run<12>("VALU + 8 ds_write_b128 + 36 ds_read_b32");
    return 0;
}
