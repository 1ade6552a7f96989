/*
 * terragan_hip.h -- C ABI of libterragan_hip.so: the MI355X (gfx950) kernels behind TERRA-GAN's
 * partial-convolution inpainting train step.
 *
 * The reference (/root/reference) is pure Python on stock ATen ops and has no native interface;
 * each entry point below names the reference arithmetic it replaces (file:line) and is what a
 * maintainer would bind (ctypes stub in INTEGRATION.md) from mvp_gan/src/models/{pconv,generator,discriminator}.py,
 * mvp_gan/src/utils/losses.py and mvp_gan/src/train.py.
 *
 * Conventions
 *   - plain C, no C++/torch types; all pointers are DEVICE pointers (fp32 unless noted);
 *   - activations are NHWC ("channels-last"): [B][H][W][C]; masks are [B][H][W] fp32 {0,1},
 *     1 = valid pixel, 0 = hole (dataset.py:37, generator.py:60-62);
 *   - conv weights are [Cout][kh][kw][Cin] (= the channels_last storage of an OIHW tensor);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never allocates,
 *     never synchronises; scratch memory is supplied by the caller (`ws`, sized by the matching
 *     tg_*_ws_bytes query);
 *   - return 0 on success, negative TG_ERR_* otherwise; tg_last_error() gives the thread-local
 *     message.  Invalid shapes are rejected on the host before any launch.
 */
#ifndef TERRAGAN_HIP_H
#define TERRAGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* tg_stream_t; /* hipStream_t */

enum { TG_OK = 0, TG_ERR_ARG = -1, TG_ERR_LAUNCH = -2, TG_ERR_WS = -3 };
enum { TG_ACT_NONE = 0, TG_ACT_RELU = 1, TG_ACT_LEAKY = 2 };
/* Arithmetic of the conv inner product.  TG_PREC_BF16: operands are rounded to bf16 inside the kernel and multiplied
 * on the bf16 MFMA with fp32 accumulation; tensors in memory, bias, mask ratio and epilogue stay fp32
 * (BASELINE config 3: bf16 compute, fp32 master weights).  Kernels without a bf16 variant ignore it. */
enum { TG_PREC_F32 = 0, TG_PREC_BF16 = 1, TG_PREC_F32_WINO4 = 2 };
/* TG_PREC_F32_WINO4: fp32 arithmetic as TG_PREC_F32, but stride-1 3x3 convolutions (forward and dgrad) with Cin % 8 == 0,
 * Cout % 64 == 0, >= 16 x 32 outputs, no input mask / row scale and tensors below 2 GB run as Winograd F(4x4,3x3) (36 multiplies per 16 outputs instead of 16 per 4): faster,
 * with 6-7x the rms (~20x the maximum: 1e-5 of the tensor's largest value) rounding error of the default F(2x2,3x3) per layer.  The train step requests it for the frozen VGG16 trunk of
 * the perceptual loss only (losses.py:31-34,79-90), whose stated tolerances it keeps; everything else ignores the hint. */

/* Geometry of one 2-D convolution (square kernel, symmetric padding). */
typedef struct TgConv {
    int32_t B, H, W, Cin;   /* input  [B][H][W][Cin]   */
    int32_t Ho, Wo, Cout;   /* output [B][Ho][Wo][Cout] */
    int32_t k, stride, pad;
    int32_t precision;      /* TG_PREC_* */
} TgConv;

int tg_version(void);
const char* tg_last_error(void);

/* ---- partial / plain convolution: implicit-GEMM on fp32 MFMA ------------------------------- */

/* y = act( (conv(x (.) in_mask, w) + bias) * ratio )
 * Replaces PConv2d.forward's input*mask -> input_conv -> output*mask_ratio (pconv.py:27-30,43)
 * when in_mask/ratio are given, and nn.Conv2d (+LeakyReLU/ReLU) otherwise
 * (generator.py:29,56; discriminator.py:11,15,22; losses.py:32 VGG trunk).
 * in_mask [B][H][W], bias [Cout], ratio [B][Ho][Wo] may be NULL. */
size_t tg_conv_fwd_ws_bytes(const TgConv* g);
int tg_conv_fwd(const TgConv* g, const float* x, const float* in_mask, const float* w,
                const float* bias, const float* ratio, int act, float slope, float* y,
                float* ws, size_t ws_bytes, tg_stream_t stream);

/* dx (+)= conv_transpose(dy, w) (.) in_mask   -- autograd of the conv above w.r.t. x.
 * dy must already carry the ratio factor.  accumulate!=0 adds into dx (skip connections).
 * ws >= tg_conv_dgrad_ws_bytes (holds the [Cin][kh][kw][Cout] transposed weights + split-K). */
size_t tg_conv_dgrad_ws_bytes(const TgConv* g);
int tg_conv_dgrad(const TgConv* g, const float* dy, const float* w, const float* in_mask,
                  float* dx, int accumulate, float* ws, size_t ws_bytes, tg_stream_t stream);
/* Same, with the backward of the activation that PRODUCED x fused into the epilogue:
 * dx = (convT(dy, w) (.) in_mask) * act'(x_act), x_act = that activation's output [B][H][W][Cin]
 * (ReLU of the VGG trunk, LeakyReLU of the discriminator's first block). */
int tg_conv_dgrad_gated(const TgConv* g, const float* dy, const float* w, const float* in_mask,
                        const float* x_act, int act, float slope, float* dx, float* ws,
                        size_t ws_bytes, tg_stream_t stream);

/* Prepared weights.  The weight rearrangements inside tg_conv_fwd / tg_conv_dgrad (Winograd transform of the stride-1 3x3
 * layers, the [Cin][kh][kw][Cout] transpose of the gather dgrads, the 3x3 x 4C regrouping of 5x5 stride-2 layers) depend
 * on the weights only: a caller that keeps them once per optimiser step (once ever for the frozen VGG trunk of
 * losses.py:31-34) saves ~45 small launches per train step.  tg_conv_wprep_bytes == 0: that (geometry, mode) runs on the
 * raw weights.  The *_p entry points equal tg_conv_fwd / tg_conv_dgrad / tg_conv_dgrad_gated (x_act != NULL) when
 * wprep == NULL; with wprep they skip the preparation and need no workspace room for it.  A prepared buffer is valid for
 * the geometry (all TgConv fields but B) and mode it was made for, until the weights change. */
enum { TG_WPREP_FWD = 0, TG_WPREP_DGRAD = 1 };
size_t tg_conv_wprep_bytes(const TgConv* g, int mode);
int tg_conv_wprep(const TgConv* g, int mode, const float* w, float* wprep, tg_stream_t stream);
/* Batched form: tg_conv_wprep_item writes a POD descriptor (tg_conv_wprep_item_bytes() bytes, HOST memory) of what
 * tg_conv_wprep(g, mode, w, wprep) would launch and returns 1 -- or returns 0 when that preparation cannot be batched (it
 * stays with tg_conv_wprep).  The caller copies the descriptors of all its layers into one DEVICE array once and has them
 * executed by ONE launch per optimiser step: tg_conv_wprep_run(items_dev, n).  The descriptors hold the w / wprep pointers:
 * they stay valid while those allocations live. */
size_t tg_conv_wprep_item_bytes(void);
int tg_conv_wprep_item(const TgConv* g, int mode, const float* w, float* wprep, void* item_out);
int tg_conv_wprep_run(const void* items_dev, int n, tg_stream_t stream);
int tg_conv_fwd_p(const TgConv* g, const float* x, const float* in_mask, const float* w, const float* wprep,
                  const float* bias, const float* ratio, int act, float slope, float* y, float* ws,
                  size_t ws_bytes, tg_stream_t stream);
/* tg_conv_fwd_p plus the 2x2 / stride-2 max-pool of its (activated) output: y [B][Ho][Wo][Cout] AND pool_y [B][Ho/2][Wo/2][Cout]
 * (Ho, Wo even).  Replaces  features[i](x) -> ReLU -> MaxPool2d(2)  of the frozen VGG16 trunk
 * (/root/reference/mvp_gan/src/utils/losses.py:31-34,79-90: torchvision vgg16().features[:16], layers 2-4 and 7-9) in one call:
 * the stride-1 3x3 fp32 Winograd kernel writes the pooled tensor from its output transform (a 2x2 output tile is a pooling
 * window); every other geometry runs tg_maxpool2_fwd on y.  Same values either way. */
int tg_conv_fwd_pool(const TgConv* g, const float* x, const float* in_mask, const float* w, const float* wprep,
                     const float* bias, const float* ratio, int act, float slope, float* y, float* pool_y, float* ws,
                     size_t ws_bytes, tg_stream_t stream);
/* conv -> ReLU -> MaxPool2d(2) where the full-resolution conv output has no other reader than the pool and the pool's backward
 * (torchvision vgg16().features[2..4] and [7..9], losses.py:31-34,79-90): writes the pooled tensor [B][Ho/2][Wo/2][Cout] and one
 * CODE byte per pooled element and channel (bits 0-1: window position 2*row + column of the maximum, the first one as in ATen;
 * bit 2: maximum > 0, i.e. the ReLU gate) -- the conv output itself is never written.  tg_maxpool2_bwd_code turns the pooled
 * gradient + code into the full-resolution gradient in front of the ReLU.  Only where the fp32 Winograd kernel takes the
 * launch in one K split: tg_conv_pool_code_supported (callers fall back to tg_conv_fwd_pool / tg_maxpool2_bwd). */
int tg_conv_pool_code_supported(const TgConv* g);
int tg_conv_fwd_pool_code(const TgConv* g, const float* x, const float* w, const float* wprep, const float* bias,
                          float* pool_y, unsigned char* code, float* ws, size_t ws_bytes, tg_stream_t stream);
int tg_maxpool2_bwd_code(const float* dout, const unsigned char* code, int B, int Ho, int Wo, int C, float* dx,
                         tg_stream_t stream);
/* BatchNorm + activation on load: the layer's input is act(BN(x)) -- x the PRE-BatchNorm output of the layer below, statistics
 * and affine parameters in `bn` -- and that tensor is never written.  For `final` (generator.py:29,56: Conv2d(64, 1, 3, 1, 1) over
 * dec1's ReLU(BN(.)) output, pconv.py:43-48): saves dec1's BatchNorm-apply pass (one read + one write of the widest activation).
 * Same rounding sequence as tg_bn_act_fwd, so results equal the two-call form bit for bit.  tg_conv_bnin_supported(g, wgrad)
 * says whether the geometry has the kernel (64 -> 1 channels, 3x3, stride 1, pad 1, W % 4 == 0, at least 4 x 16 pixels);
 * the two calls fail otherwise (the caller then materialises the activation with tg_bn_act_fwd). */
typedef struct TgBnAct {
    const float *mean, *rstd, *gamma, *beta;   /* [Cin] each */
    int32_t act;                               /* TG_ACT_* */
    float slope;
} TgBnAct;
int tg_conv_bnin_supported(const TgConv* g, int wgrad);
int tg_conv_fwd_bnin(const TgConv* g, const float* x, const TgBnAct* bn, const float* w, const float* bias, int act,
                     float slope, float* y, float* ws, size_t ws_bytes, tg_stream_t stream);
int tg_conv_wgrad_bnin(const TgConv* g, const float* x, const TgBnAct* bn, const float* dy, float* dw, float* db, float* ws,
                       size_t ws_bytes, tg_stream_t stream);
int tg_conv_dgrad_p(const TgConv* g, const float* dy, const float* w, const float* wprep, const float* in_mask,
                    const float* x_act, int act, float slope, float* dx, int accumulate, float* ws,
                    size_t ws_bytes, tg_stream_t stream);

/* Leave `cus` of the 256 CUs free in the launches that otherwise occupy every CU with one long-running workgroup (the
 * Winograd kernels): data-parallel runs set this so that RCCL's kernels on the communication stream can be scheduled
 * while a convolution is running (0 = default, single-GPU). */
int tg_set_cu_reserve(int cus);
/* mode 1: the persistent Winograd launches with two or more work items per workgroup hand out the items behind each
 * workgroup's first one from per-XCD queues (one atomic per item) instead of walking a fixed list, so that a workgroup whose
 * CU is held by another stream's long-running kernel (RCCL's reductions in a data-parallel run) delays one item, not its
 * whole list (profiles/r03_ws_contention.txt: +90 % -> +8 % with 4-16 CUs held).  mode 2: every such launch (tests).
 * mode 0 (default, single GPU): static lists -- 1 % faster on an uncontended chip.  Results are bitwise equal in all modes. */
int tg_set_work_stealing(int mode);

/* dw[Cout][k][k][Cin] = sum_pixels dy (x) (x (.) in_mask);  db[Cout] = sum_pixels dy (db may be NULL).
 * Deterministic: split-K partial slabs in ws, reduced in a fixed order. */
size_t tg_conv_wgrad_ws_bytes(const TgConv* g);
int tg_conv_wgrad(const TgConv* g, const float* x, const float* in_mask, const float* dy,
                  float* dw, float* db, float* ws, size_t ws_bytes, tg_stream_t stream);

/* w3 [Cout][k][k][Cin] -> w1 [Cout][k][k][1] = sum over Cin: the VGG first conv sees the grey
 * image repeated x3 (losses.py:79-80), i.e. a 1-channel conv with the channel-summed kernel. */
int tg_fold_cin(const float* w3, int cout, int taps, int cin, float* w1, tg_stream_t stream);

/* ---- mask path (exact integer arithmetic in fp32 storage) ---------------------------------- */

/* S = window sum of mask; mask_out = [S>0]; ratio = k*k/(S+1e-8)*[S>0]   (pconv.py:33-40). */
int tg_mask_update(const float* mask, int B, int H, int W, int k, int stride, int pad, int Ho,
                   int Wo, float* mask_out, float* ratio, tg_stream_t stream);
/* out = max(nearest_up2(up_mask) zero-padded to [H][W], skip_mask)  (generator.py:51-54,68-74). */
int tg_mask_up_merge(const float* up_mask, const float* skip_mask, int B, int h, int w, int H,
                     int W, float* out, tg_stream_t stream);

/* The whole mask pyramid of one generator forward (14 x tg_mask_update + 7 x tg_mask_up_merge, which depend on the input
 * mask only) as ONE launch: `op[i]` is applied to every image after op[i-1] (one workgroup per image walks the levels).
 * kind 0 = tg_mask_update(in[B][H][W]; k, stride, pad) -> out (mask) [B][Ho][Wo], out2 (ratio);
 * kind 1 = tg_mask_up_merge(in = up_mask [B][H][W], in2 = skip_mask [B][Ho][Wo]) -> out [B][Ho][Wo].
 * Results are bit-identical to the per-level entry points. */
#define TG_MASK_PYRAMID_MAX 24
typedef struct TgMaskOp {
    int kind, H, W, Ho, Wo, k, stride, pad;
    const float* in;
    const float* in2;
    float* out;
    float* out2;
} TgMaskOp;
typedef struct TgMaskPyramid {
    int nops, _pad;
    TgMaskOp op[TG_MASK_PYRAMID_MAX];
} TgMaskPyramid;
int tg_mask_pyramid(const TgMaskPyramid* pm, int B, tg_stream_t stream);

/* ---- BatchNorm (+ReLU / LeakyReLU) ---------------------------------------------------------- */

/* Training-mode batch statistics of y[rows][C] (biased var, eps inside rstd) and the running
 * update with `momentum` (unbiased var) -- nn.BatchNorm2d, pconv.py:21,47; discriminator.py:13.
 * running_* and num_batches_tracked (int64) may be NULL.  ws >= tg_bn_ws_bytes(rows, C). */
size_t tg_bn_ws_bytes(int64_t rows, int C);
int tg_bn_stats(const float* y, int64_t rows, int C, float eps, float momentum, float* save_mean,
                float* save_rstd, float* running_mean, float* running_var,
                int64_t* num_batches_tracked, float* ws, size_t ws_bytes, tg_stream_t stream);
/* tg_bn_stats followed (when out != NULL) by tg_bn_act_fwd as one call: one launch for small maps (few rows: the bottleneck
 * levels of the U-Net, where the separate launches are pure latency), the same kernels as the two calls otherwise.
 * ws as for tg_bn_stats. */
int tg_bn_fwd(const float* y, int64_t rows, int C, float eps, float momentum, const float* gamma,
              const float* beta, int act, float slope, float* save_mean, float* save_rstd,
              float* running_mean, float* running_var, int64_t* num_batches_tracked, float* out,
              float* ws, size_t ws_bytes, tg_stream_t stream);
/* eval mode: mean = running_mean, rstd = 1/sqrt(running_var + eps). */
int tg_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps,
                     float* mean, float* rstd, tg_stream_t stream);
/* out = act( (y-mean)*rstd*gamma + beta )   (pconv.py:47-48; discriminator.py:13-14). */
int tg_bn_act_fwd(const float* y, int64_t rows, int C, const float* mean, const float* rstd,
                  const float* gamma, const float* beta, int act, float slope, float* out,
                  tg_stream_t stream);
/* Backward of bn_act_fwd in training mode.  dy = dBN(dout (.) act'(.)) [* ratio[row]], dgamma,
 * dbeta.  ratio may be NULL.  dy may alias dout.  dbias (may be NULL) receives sum_rows dy, i.e. the
 * gradient of the bias of the convolution feeding this BatchNorm, obtained in closed form from the
 * same reduction pass (no extra read of dy). */
int tg_bn_act_bwd(const float* dout, const float* y, int64_t rows, int C, const float* mean,
                  const float* rstd, const float* gamma, const float* beta, int act, float slope,
                  const float* ratio, float* dy, float* dgamma, float* dbeta, float* dbias, float* ws,
                  size_t ws_bytes, tg_stream_t stream);
/* The same when the incoming gradient is the INPUT gradient of a C -> 1 channel 3x3 / stride-1 / pad-1 convolution (dec1's
 * BatchNorm under `final`, generator.py:29,56 + pconv.py:43-48):  dout[b][y][x][c] = sum_{ky,kx} dz[b][y+1-ky][x+1-kx] * w[ky][kx][c]
 * (w = that convolution's weight, [1][3][3][C]) is recomputed from the 1-channel dz inside both passes instead of being written by
 * tg_conv_dgrad and read twice.  dy [B][H][W][C] is written (nothing is consumed in place).  tg_bn_bwd_conv1_supported(rows, C):
 * C % 4 == 0 and more rows than the one-launch small-map form takes. */
int tg_bn_bwd_conv1_supported(int64_t rows, int C);
size_t tg_bn_conv1_ws_bytes(int64_t rows, int C);
int tg_bn_act_bwd_conv1(const float* dz, const float* w, int B, int H, int W, const float* y, int C, const float* mean,
                        const float* rstd, const float* gamma, const float* beta, int act, float slope, const float* ratio,
                        float* dy, float* dgamma, float* dbeta, float* dbias, float* ws, size_t ws_bytes, tg_stream_t stream);
/* din = dout * act'(out) [* ratio[row]] for a conv epilogue activation (out = post-activation).
 * din may alias dout. */
int tg_act_bwd(const float* dout, const float* out, int64_t rows, int C, int act, float slope,
               const float* ratio, float* din, tg_stream_t stream);

/* ---- decoder plumbing: bilinear x2 upsample (+) channel concat ------------------------------- */

/* out[B][H][W][Cu+Cs] = cat( pad(bilinear_up2(up[B][h][w][Cu])), skip[B][H][W][Cs] ) (.) out_mask
 * (generator.py:50,52,67,70,73; align_corners=False; skip may be NULL with Cs=0).  out_mask
 * [B][H][W] (NULL = 1) is the merged decoder mask: writing the concat already multiplied by it is the
 * `input * mask` of the consuming PConv2d (pconv.py:27), so that conv and its wgrad read it unmasked. */
int tg_upcat_fwd(const float* up, const float* skip, const float* out_mask, int B, int h, int w,
                 int Cu, int H, int W, int Cs, float* out, tg_stream_t stream);
/* The same with `up` = the PRE-BatchNorm output of the decoder layer below: the interpolation runs over act(BN(up)) formed on
 * load (TgBnAct: see tg_conv_fwd_bnin above), so that layer's activation -- read by nothing else
 * (generator.py:66-76 consumes it once, through F.interpolate) -- is never written.  Exact x2 geometries only
 * (H == 2h, W == 2w, Cu % 4 == 0, Cs % 4 == 0): tg_upcat_bn_supported; same bits as tg_bn_act_fwd + tg_upcat_fwd. */
int tg_upcat_bn_supported(int B, int h, int w, int Cu, int H, int W, int Cs);
int tg_upcat_fwd_bn(const float* up, const TgBnAct* bn, const float* skip, const float* out_mask, int B, int h, int w,
                    int Cu, int H, int W, int Cs, float* out, tg_stream_t stream);
/* adjoint: dup[B][h][w][Cu] = bilinear_up2^T(dout[..., :Cu]);  dskip = dout[..., Cu:]. */
int tg_upcat_bwd(const float* dout, int B, int h, int w, int Cu, int H, int W, int Cs, float* dup,
                 float* dskip, tg_stream_t stream);

/* ---- generator head --------------------------------------------------------------------------- */

/* out = sigmoid(logits)*(1-mask) + x*mask   (generator.py:57-62), n = B*H*W. */
int tg_sigmoid_composite_fwd(const float* logits, const float* x, const float* mask, int64_t n,
                             float* out, tg_stream_t stream);
/* dlogits = dout*(1-mask)*s*(1-s);  dx (may be NULL) = dout*mask. */
int tg_sigmoid_composite_bwd(const float* dout, const float* logits, const float* mask, int64_t n,
                             float* dlogits, float* dx, tg_stream_t stream);

/* ---- VGG trunk helpers (losses.py:31-34,79-90) ------------------------------------------------ */
int tg_maxpool2_fwd(const float* x, int B, int H, int W, int C, float* out, tg_stream_t stream);
/* relu_gate != 0: x is a ReLU output and its backward is fused (no gradient where the max is 0). */
int tg_maxpool2_bwd(const float* dout, const float* x, int B, int H, int W, int C, int relu_gate,
                    float* dx, tg_stream_t stream);

/* ---- losses ------------------------------------------------------------------------------------ */

/* Pixel-space part of InpaintingLoss.forward (losses.py:73,98-100,118-127,404-416) in one pass
 * family over (pred, target, mask) [B][H][W]:
 *   out[0] = L1 mean, out[1] = TV(pred*(1-mask)) (B divided twice, as the reference),
 *   out[2] = boundary loss (3x3 morphological-gradient band; 0 if the band is empty or the value
 *            is NaN/Inf), out[3] = sum(band), out[4] = l1 + w_tv*tv + w_bnd*boundary.
 * l1_weight (NULL = 1) weights |pred-target| inside the L1 mean: HumanGuidedLoss's human term
 * L1(pred*h, target*h) with h in {0,1} (losses.py:173-176).
 * If dpred != NULL also writes d out[4] / d pred * (*gscale) (gscale NULL = 1), accumulating into
 * dpred when accumulate != 0.  No host synchronisation.  ws >= tg_pixel_loss_ws_bytes(). */
size_t tg_pixel_loss_ws_bytes(int B, int H, int W);
int tg_pixel_losses(const float* pred, const float* target, const float* mask,
                    const float* l1_weight, int B, int H, int W,
                    float w_l1, float w_tv, float w_bnd, float bnd_eps, const float* gscale,
                    float* out5, float* dpred, int accumulate, float* ws, size_t ws_bytes,
                    tg_stream_t stream);

/* out[0] = mean |a-b| over n;  if da != NULL: da = coef * (*gscale) * sign(a-b)/n
 * (nn.L1Loss on VGG features, losses.py:86-89). */
size_t tg_reduce_ws_bytes(int64_t n);
int tg_l1_mean(const float* a, const float* b, int64_t n, float coef, const float* gscale,
               float* out1, float* da, float* ws, size_t ws_bytes, tg_stream_t stream);
/* The same with `a` a ReLU OUTPUT and the gradient taken in front of that ReLU: da = (a > 0) ? coef*g*sign(a-b)/n : 0 -- the
 * perceptual term's L1 over relu3_3 features (losses.py:86-89) and the backward of features[15] = ReLU(inplace) in one pass. */
int tg_l1_mean_relu(const float* a, const float* b, int64_t n, float coef, const float* gscale, float* out1, float* da,
                    float* ws, size_t ws_bytes, tg_stream_t stream);

/* nn.BCEWithLogitsLoss (mean) against the constant `target` (train.py:115,203,215-216):
 * out[0] = loss; if dz != NULL: dz = coef * (*gscale) * (sigmoid(z)-target)/n. */
int tg_bce_logits(const float* z, int64_t n, float target, float coef, const float* gscale,
                  float* out1, float* dz, float* ws, size_t ws_bytes, tg_stream_t stream);

/* ---- optimiser / misc ---------------------------------------------------------------------------- */

/* torch.optim.Adam defaults (main_pipeline.py:214-221; train.py:139-147): in-place update of
 * p, m (exp_avg), v (exp_avg_sq) from g*grad_scale with bias correction for `step` (1-based). */
/* lr/betas/eps are doubles: they are rounded to fp32 exactly where torch.optim.Adam rounds its Python floats. */
int tg_adam(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
            double beta2, double eps, int step, float grad_scale, tg_stream_t stream);

/* Multi-tensor form: ONE launch over a device-resident table of segments (whole parameter tensors) and a
 * device-resident work list of int32 pairs (segment index, chunk index); chunk c of a segment covers elements
 * [c*chunk_elems, min(n, (c+1)*chunk_elems)).  All segments share lr/betas/eps/step. */
typedef struct TgAdamSeg {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
} TgAdamSeg;
int tg_adam_multi(const TgAdamSeg* segs_dev, const int32_t* work_dev, int nwork, int chunk_elems,
                  double lr, double beta1, double beta2, double eps, int step, float grad_scale,
                  tg_stream_t stream);

/* The same launch with the two per-step scalars read from DEVICE memory -- scal_dev[0] = lr / (1 - beta1^step),
 * scal_dev[1] = sqrt(1 - beta2^step), exactly the two floats tg_adam_multi derives from (lr, betas, step) and which
 * tg_adam_scalars writes to HOST memory -- so that a train step captured in a hipGraph can be replayed: the host refreshes
 * the two floats (one small copy) before each replay, every other kernel argument of a step is step-invariant. */
int tg_adam_scalars(double lr, double beta1, double beta2, int step, float* out2_host);
/* dst_dev[0..n) = vals_host[0..n), n <= 16; the values travel as kernel arguments, so vals_host may be reused at once. */
int tg_write_floats(float* dst_dev, int n, const float* vals_host, tg_stream_t stream);
int tg_adam_multi_s(const TgAdamSeg* segs_dev, const int32_t* work_dev, int nwork, int chunk_elems,
                    double beta1, double beta2, double eps, const float* scal_dev, float grad_scale,
                    tg_stream_t stream);

/* y = a*x + b*y elementwise (gradient accumulation / scaling glue). */
int tg_axpby(const float* x, float a, float b, float* y, int64_t n, tg_stream_t stream);
/* out = a*x + b*y of two device scalars/vectors into a third (loss totals). */
int tg_lincomb(const float* x, float a, const float* y, float b, float* out, int64_t n,
               tg_stream_t stream);
/* out = a*b elementwise (masked_imgs = real_imgs * masks, train.py:181). */
int tg_mul(const float* a, const float* b, float* out, int64_t n, tg_stream_t stream);
/* out = a*b and a_copy = a from one read of a: masked_imgs (train.py:181) plus real_imgs placed behind the generated batch
 * in the stacked [gen; real] buffer the loss trunk (losses.py:79-88) and the discriminator passes (train.py:202,211) read. */
int tg_mul_keep(const float* a, const float* b, float* out, float* a_copy, int64_t n, tg_stream_t stream);
/* Re-apply the BatchNorm running-stat momentum update from saved batch statistics
 * (mean, rstd as written by tg_bn_stats): used when a forward pass is provably identical to one
 * already computed (D(gen) and D(gen.detach()), train.py:202,212) and only its running-stat side
 * effect remains to be reproduced. */
/* ---- BatchNorm over `groups` passes stacked along the rows: statistics PER PASS, one set of launches -------------------------
 * The train step stacks D(fake) and D(real) (train.py:202,211) into one forward / one backward; each pass keeps its own batch
 * statistics (discriminator.py:13: nn.BatchNorm2d in train mode, once per call).  y / out / dout / dy: [groups*rows_per_group][C],
 * mean / rstd: [groups][C].  Bit-identical to `groups` separate tg_bn_stats + tg_bn_act_fwd (resp. tg_bn_act_bwd) calls whose
 * parameter gradients are added in pass order.  C % 4 == 0.  No running-statistics side effect: tg_bn_running_update_multi. */
size_t tg_bn_grouped_ws_bytes(int64_t rows_per_group, int groups, int C);
int tg_bn_fwd_grouped(const float* y, int64_t rows_per_group, int groups, int C, float eps, const float* gamma,
                      const float* beta, int act, float slope, float* save_mean, float* save_rstd, float* out,
                      float* ws, size_t ws_bytes, tg_stream_t stream);
int tg_bn_act_bwd_grouped(const float* dout, const float* y, int64_t rows_per_group, int groups, int C,
                          const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                          float slope, float* dy, float* dgamma, float* dbeta, float* dbias, float* ws,
                          size_t ws_bytes, tg_stream_t stream);
/* tg_bn_running_update for passes order[0], order[1], ... (indices into mean / rstd [groups][C]) applied one after the other:
 * the reference updates model.N's buffers in the order D(fake), D(real), D(fake.detach()) (train.py:202,211,212). */
int tg_bn_running_update_multi(const float* save_mean, const float* save_rstd, int64_t rows_per_group, int C,
                               float eps, float momentum, const int* order, int norder, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, tg_stream_t stream);
int tg_bn_running_update(const float* save_mean, const float* save_rstd, int64_t rows, int C,
                         float eps, float momentum, float* running_mean, float* running_var,
                         int64_t* num_batches_tracked, tg_stream_t stream);
/* [B][C][H][W] <-> [B][H][W][C] transposes (API boundary only; C==1 tensors need none). */
int tg_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float* y, tg_stream_t stream);
int tg_nhwc_to_nchw(const float* x, int B, int C, int H, int W, float* y, tg_stream_t stream);

/* ---- measurement hooks (bench.py roofline figures; no reference counterpart) ------------------- */

/* Quality metrics the reference logs per batch / validation pass, in ONE pass over [imgs][H][W] fp32 tensors
 * (imgs = B*C) with no host synchronisation:
 *   out[0] mse   out[1] psnr = 20 log10(1/sqrt(mse)) (inf when mse == 0)   out[2] ssim (11x11 avg_pool2d windows, zero
 *   padding, divisor 121, C1 = 0.01^2, C2 = 0.03^2)   out[3] l1   out[4] l2 = sqrt(mse)
 *     -- utils/experiment_tracking.py:176-231 (= MaskEvaluator._calculate_psnr/_ssim, evaluation/metrics.py:47-76)
 *   out[5] boundary_mse = mean over ALL elements of ((pred-target)*band)^2, band = clamp(maxpool3(m) - (1 - maxpool3(1-m)))
 *   out[6] boundary_psnr = 10 log10(1 / (boundary_mse + 1e-6))
 *   out[7] boundary_gradient_diff = | (mean|dy pred| + mean|dx pred|) - (mean|dy target| + mean|dx target|) |
 *   out[8] sum(band); out[5..7] are 0 when sum(band) < 1e-6
 *     -- calculate_boundary_quality, mvp_gan/src/evaluation/metrics.py:79-133
 * ws: 8-byte aligned, >= tg_quality_metrics_ws_bytes(). */
size_t tg_quality_metrics_ws_bytes(int64_t imgs, int H, int W);
int tg_quality_metrics(const float* pred, const float* target, const float* mask, int64_t imgs, int H, int W,
                       float* out9, float* ws, size_t ws_bytes, tg_stream_t stream);

/* Pre-decoded uint8 tile shards -> fp32 tiles on the device: img_f32 = img_u8 / 255 (IEEE fp32 division),
 * mask_f32 = mask_u8 > 0 -- mvp_gan/src/utils/dataset.py:35-37 (ToTensor scaling, binarise AFTER the resize).
 * Either input may be NULL.  n elements; pointers 16-byte aligned. */
int tg_u8_to_tiles(const uint8_t* img_u8, const uint8_t* mask_u8, int64_t n, float* img_f32, float* mask_f32,
                   tg_stream_t stream);

/* When enabled, every launch of the MFMA conv kernels is bracketed by hipEvents on its own launch
 * stream and tagged with its algorithmic FLOPs and bytes.  kind: 0 = fwd/dgrad implicit GEMM,
 * 1 = wgrad.  tg_prof_summary synchronises those events (host-blocking: call it outside any timed
 * region), returns the totals for `kind` and consumes its records. */
int tg_prof_enable(int on);
int tg_prof_summary(int kind, double* total_ms, int64_t* launches, double* flops, double* bytes);
/* One CSV row per recorded launch (kind,cfg,M,N,K,C,splits,ms,gflop,alg_mb,tag); records are kept. */
int tg_prof_dump(const char* path);
/* Label (<= 31 chars, e.g. "dec1.fwd") attached to the launches recorded after it on the calling thread. */
int tg_prof_tag(const char* tag);

#ifdef __cplusplus
}
#endif
#endif /* TERRAGAN_HIP_H */
