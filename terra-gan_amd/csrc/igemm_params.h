// Parameter blocks shared by the MFMA conv kernels (igemm.hip) and the small-channel kernels (smallconv.hip).
#pragma once
#include "common.h"

// BatchNorm + activation applied to the SOURCE tensor as it is staged (tg_conv_fwd_bnin / tg_conv_wgrad_bnin): the layer's input
// is act(BN(src)) without that tensor ever being written.  mean == nullptr: none.
struct BnIn {
    const float *mean, *rstd, *gamma, *beta;
    int act;
    float slope;
};
struct IGemmParams {
    const float* src;       // A source, NHWC [B][IH][IW][C]
    const float* amask;     // optional [B][IH][IW]: A row scale at the SOURCE pixel (x (.) mask)
    const float* wmat;      // B matrix [N][Kfull], K-contiguous, K index = tapidx*C + c
    const float* bias;      // optional [N]
    const float* rowscale;  // optional, indexed by DESTINATION pixel (ratio / dgrad mask)
    float* dst;             // NHWC [B][DH][DW][N]
    float* ws;              // split-K slabs [splits][M][N] when splits > 1
    int B, IH, IW, C;
    int OH, OW, N, M;       // output grid of this launch, M = B*OH*OW
    int DH, DW, ds, dy0, dx0;           // grid point (oy,ox) -> dst pixel (oy*ds+dy0, ox*ds+dx0)
    int TH, TW, ss, tstep, sy0, sx0;    // tap (ty,tx) -> src pixel (oy*ss+sy0+ty*tstep, ...)
    int KW, kstep, ky0, kx0;            // tap (ty,tx) -> weight tap (ky0+ty*kstep)*KW + kx0+tx*kstep
    int Kfull, Ktot;        // wmat row length; K elements walked by this launch (TH*TW*C)
    int nchunks, T;         // ceil(C/32); number of 32-deep K steps
    int splits, steps_per_split;
    int act;
    float slope;
    int accumulate;
    int bf16;               // operands rounded to bf16 at LDS staging, v_mfma_f32_32x32x16_bf16 (patch kernel only)
    const float* gate;      // optional [dst pixels][N]: result *= act'(gate) (fused activation backward)
    int gate_act;
    float gate_slope;
    // Winograd path (wino.inc): raw weights with the element strides of (output row n, contraction k, tap), and room
    // for the transformed weights; wino_u == nullptr disables it
    const float* w_raw;
    long w_sn, w_sk, w_stap;
    float* wino_u;
    int wino4;              // TG_PREC_F32_WINO4: Winograd F(4x4,3x3) where the geometry allows (wino44.inc)
    int wino_ready;         // wino_u already holds the transformed weights (prepared by tg_conv_wprep): skip the transform
    // 2x2 / stride-2 max-pool of the (activated) output, written next to dst from the output transform (tg_conv_fwd_pool):
    // [B][OH/2][OW/2][N].  A launcher that writes it sets pool_done; otherwise the caller runs the pool kernel on dst.
    float* pool_dst;
    int pool_done;
    // tg_conv_fwd_pool_code: with the pooled tensor a BYTE per pooled element and channel -- bits 0-1 the window position of the
    // maximum (2*row + column, first maximum wins as in ATen), bit 2 "maximum > 0" (the ReLU gate) -- which is all the pool's
    // backward needs; pool_only: dst itself is NOT written (its only readers were the pool and that backward)
    unsigned char* pool_code;
    int pool_only;
    BnIn in_bn;             // (the 64 -> 1 channel LDS-patch kernel only: `final`, smallconv.hip)
};
__device__ __forceinline__ float gate_factor(const IGemmParams& p, size_t idx) {
    const float gv = p.gate[idx];
    return gv > 0.f ? 1.f : (p.gate_act == TG_ACT_LEAKY ? p.gate_slope : 0.f);
}

// Up to 4 independent problems in ONE launch (the parity classes of a stride-2 dgrad)
struct IGemmMulti {
    IGemmParams c[4];
    int zbeg[5];            // igemm_multi_kernel: class i owns blockIdx.z in [zbeg[i], zbeg[i + 1]) -- its own split count (c[i].splits)
};

struct WgradParams {
    const float* x;
    const float* amask;
    const float* dy;
    float* out;  // [splits][Cout][Ktot]
    int B, H, W, C, Ho, Wo, Cout, k, stride, pad;
    int Mpix, Ktot, T, splits, steps_per_split;
    int nx, ny;   // N' tiles, Cout tiles (grid is launched flat: nx*ny*splits workgroups)
    int rowseg;   // Wo % 32 == 0: every 32-pixel K step lies inside one output row (cheap gather addressing)
    BnIn in_bn;   // (to1wgrad64_lds_kernel only)
};

// smallconv.hip: bandwidth-bound special cases that would waste >95% of an MFMA tile
bool smallconv_fwd_applies(const IGemmParams& p);             // C == 1 -> N%64 == 0, or N == 1 <- C%64 == 0
int smallconv_fwd_launch(const IGemmParams& p, hipStream_t s);
bool smallconv_to1_multi_applies(const IGemmParams* cls, int ncls);   // 64 -> 1 channel, the four 2x2-tap classes of a 4x4 stride-2 dgrad
int smallconv_to1_multi_launch(const IGemmParams* cls, int ncls, hipStream_t s);
bool smallconv_wgrad_applies(const WgradParams& p);
bool smallconv_bnin_fwd_ok(const IGemmParams& p);             // launches that can take IGemmParams::in_bn / WgradParams::in_bn
bool smallconv_bnin_wgrad_ok(const WgradParams& p);
size_t smallconv_wgrad_ws_floats(const WgradParams& p);
int smallconv_wgrad_launch(const WgradParams& p, float* dw, float* ws, hipStream_t s, float* db = nullptr, int* db_done = nullptr);
