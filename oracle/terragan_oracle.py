"""CPU oracle for the TERRA-GAN partial-conv inpainting train step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``terra-gan_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do, and there only as the checker / reported baseline.

This is a functional, fp32, PyTorch-CPU restatement of the arithmetic of the
reference hot path (it is *not* a copy of the reference modules: parameters
live in flat dicts keyed like the reference state-dicts, and every stage is a
plain function).  Each function cites the reference file:line it follows
(paths relative to /root/reference).

Pinning: the reference has no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned by fixtures produced from the
reference's own modules, imported by file path in the build container
(``tests/golden/make_golden.py``, committed with its outputs) and checked in
``tests/test_oracle_golden.py``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

# (name, cin, cout, k, stride, pad) in construction order -- generator.py:13-28
G_LAYERS: List[Tuple[str, int, int, int, int, int]] = [
    ("enc1", 1, 64, 7, 2, 3),
    ("enc2", 64, 128, 5, 2, 2),
    ("enc3", 128, 256, 5, 2, 2),
    ("enc4", 256, 512, 3, 2, 1),
    ("enc5", 512, 512, 3, 2, 1),
    ("enc6", 512, 512, 3, 2, 1),
    ("enc7", 512, 512, 3, 2, 1),
    ("dec7", 1024, 512, 3, 1, 1),
    ("dec6", 1024, 512, 3, 1, 1),
    ("dec5", 1024, 512, 3, 1, 1),
    ("dec4", 768, 256, 3, 1, 1),
    ("dec3", 384, 128, 3, 1, 1),
    ("dec2", 192, 64, 3, 1, 1),
    ("dec1", 64, 64, 3, 1, 1),
]
G_SPEC = {n: (ci, co, k, s, p) for n, ci, co, k, s, p in G_LAYERS}

# (sequential index, cin, cout, k, stride, pad, bn index or None, leaky) -- discriminator.py:17-23
D_LAYERS = [
    (0, None, 64, 4, 2, 1, None, True),
    (2, 64, 128, 4, 2, 1, 3, True),
    (5, 128, 256, 4, 2, 1, 6, True),
    (8, 256, 512, 4, 2, 1, 9, True),
    (11, 512, 1, 4, 1, 1, None, False),
]

# torchvision VGG16 "D" configuration; features[:16] ends at relu3_3 (losses.py:31-32)
VGG16_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
VGG_TRUNK = [(0, 3, 64), (2, 64, 64), "M", (5, 64, 128), (7, 128, 128), "M",
             (10, 128, 256), (12, 256, 256), (14, 256, 256)]

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------
# seeded initialisation, bit-identical to constructing the reference modules
# ----------------------------------------------------------------------------
def _conv_init(cout: int, cin: int, k: int, bias: bool = True):
    """nn.Conv2d.reset_parameters: kaiming_uniform(a=sqrt 5) then bias U(+-1/sqrt(fan_in))."""
    w = torch.empty(cout, cin, k, k)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    b = None
    if bias:
        bound = 1.0 / math.sqrt(cin * k * k)
        b = torch.empty(cout)
        torch.nn.init.uniform_(b, -bound, bound)
    return w, b


def _bn_init(p: Params, prefix: str, c: int) -> None:
    p[prefix + ".weight"] = torch.ones(c)
    p[prefix + ".bias"] = torch.zeros(c)
    p[prefix + ".running_mean"] = torch.zeros(c)
    p[prefix + ".running_var"] = torch.ones(c)
    p[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def init_pconv(p: Params, name: str, cin: int, cout: int, k: int) -> None:
    """RNG order of one PConv2d(): input_conv, mask_conv (discarded draw), bn -- pconv.py:9-21."""
    w, b = _conv_init(cout, cin, k)
    p[f"{name}.input_conv.weight"], p[f"{name}.input_conv.bias"] = w, b
    _conv_init(1, 1, k, bias=False)          # mask_conv draw, overwritten by ones (pconv.py:11-14)
    p[f"{name}.mask_conv.weight"] = torch.ones(1, 1, k, k)
    _bn_init(p, f"{name}.bn", cout)


def init_generator() -> Params:
    """RNG order of PConvUNet() -- generator.py:9-29, pconv.py:7-21."""
    p: Params = {}
    for name, cin, cout, k, _s, _p in G_LAYERS:
        init_pconv(p, name, cin, cout, k)
    p["final.weight"], p["final.bias"] = _conv_init(1, 64, 3)
    return p


def init_discriminator(input_channels: int = 1) -> Params:
    """RNG order of Discriminator() -- discriminator.py:10-23."""
    p: Params = {}
    for idx, cin, cout, k, _s, _p, bn, _l in D_LAYERS:
        cin = input_channels if cin is None else cin
        p[f"model.{idx}.weight"], p[f"model.{idx}.bias"] = _conv_init(cout, cin, k)
        if bn is not None:
            _bn_init(p, f"model.{bn}", cout)
    return p


def init_vgg_standin() -> Params:
    """Deterministic stand-in for torchvision vgg16().features (ImageNet weights are not
    fetchable offline): all 13 convs drawn with nn.Conv2d's default init, in
    order; only features[:16] are kept (losses.py:31-32)."""
    p: Params = {}
    idx, cin = 0, 3
    for v in VGG16_CFG:
        if v == "M":
            idx += 1
            continue
        w, b = _conv_init(v, cin, 3)
        if idx < 16:
            p[f"{idx}.weight"], p[f"{idx}.bias"] = w, b
        cin = v
        idx += 2
    return p


def trainable(p: Params) -> List[str]:
    """Keys Adam updates: everything but BN buffers and the frozen mask_conv ones (pconv.py:15-16)."""
    return [k for k in p if not (k.endswith("running_mean") or k.endswith("running_var")
                                 or k.endswith("num_batches_tracked") or ".mask_conv." in k)]


# ----------------------------------------------------------------------------
# model arithmetic
# ----------------------------------------------------------------------------
def batch_norm(x: Tensor, p: Params, prefix: str, training: bool) -> Tensor:
    """nn.BatchNorm2d.forward (eps 1e-5, momentum 0.1): batch statistics + running-stat update
    in training, running statistics in eval.  Calls the same ATen batch_norm the reference's
    nn.BatchNorm2d calls (pconv.py:21,47; discriminator.py:13), so the ill-conditioned tiny-batch
    bottleneck layers (2-4 values per channel) round identically."""
    if training:
        if x.numel() // x.shape[1] <= 1:
            raise ValueError("Expected more than 1 value per channel when training")
        p[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                        p[prefix + ".weight"], p[prefix + ".bias"], training, BN_MOMENTUM, BN_EPS)


def mask_update(mask: Tensor, k: int, stride: int, pad: int) -> Tuple[Tensor, Tensor]:
    """pconv.py:33-40: S = all-ones conv of the 1-channel mask; m' = [S>0];
    ratio = k*k/(S+1e-8)*[S>0]."""
    ones = torch.ones(1, 1, k, k, dtype=mask.dtype)
    s = F.conv2d(mask, ones, None, stride, pad)
    out_mask = (s > 0).float()
    ratio = (k * k) / (s + 1e-8) * out_mask
    return out_mask, ratio


def pconv(x: Tensor, mask: Tensor, p: Params, name: str, training: bool = True,
          batch_norm_on: bool = True, spec=None) -> Tuple[Tensor, Tensor]:
    """PConv2d.forward -- pconv.py:25-50.  Bias is renormalised with the conv output (:30,43).
    `spec` = (cin, cout, k, stride, pad) overrides the generator table for stand-alone layers."""
    _ci, _co, k, s, pad = spec if spec is not None else G_SPEC[name]
    y = F.conv2d(x * mask, p[f"{name}.input_conv.weight"], p[f"{name}.input_conv.bias"], s, pad)
    with torch.no_grad():
        out_mask, ratio = mask_update(mask, k, s, pad)
    y = y * ratio
    if batch_norm_on:
        y = batch_norm(y, p, f"{name}.bn", training)
    return F.relu(y), out_mask


def pad_to_match(x: Tensor, ref: Tensor) -> Tensor:
    """generator.py:78-84."""
    dy, dx = ref.shape[2] - x.shape[2], ref.shape[3] - x.shape[3]
    if dy == 0 and dx == 0:
        return x
    return F.pad(x, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])


def decode_step(up: Tensor, up_mask: Tensor, skip: Tensor, skip_mask: Tensor, p: Params,
                name: str, training: bool) -> Tuple[Tensor, Tensor]:
    """generator.py:66-76."""
    up = F.interpolate(up, scale_factor=2, mode="bilinear", align_corners=False)
    up_mask = F.interpolate(up_mask, scale_factor=2, mode="nearest")
    up, up_mask = pad_to_match(up, skip), pad_to_match(up_mask, skip_mask)
    return pconv(torch.cat([up, skip], 1), torch.maximum(up_mask, skip_mask), p, name, training)


def generator_forward(p: Params, x: Tensor, mask: Tensor, training: bool = True,
                      taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """PConvUNet.forward -- generator.py:31-64 (x is the already-masked image).
    `taps` (tests only): collects every layer's output activation (graph tensors: their gradients are the
    activation gradients of the backward chain, tests/golden/make_golden.py steps_chain)."""
    e, m = [x], [mask]
    for i in range(1, 8):
        a, b = pconv(e[-1], m[-1], p, f"enc{i}", training)
        if taps is not None:
            taps[f"enc{i}"] = a
        e.append(a)
        m.append(b)
    d, dm = e[7], m[7]
    for lvl, name in zip(range(6, 0, -1), ["dec7", "dec6", "dec5", "dec4", "dec3", "dec2"]):
        d, dm = decode_step(d, dm, e[lvl], m[lvl], p, name, training)
        if taps is not None:
            taps[name] = d
    up = pad_to_match(F.interpolate(d, scale_factor=2, mode="bilinear", align_corners=False), x)
    upm = pad_to_match(F.interpolate(dm, scale_factor=2, mode="nearest"), mask)
    d0, _ = pconv(up, torch.maximum(upm, mask), p, "dec1", training)
    logits = F.conv2d(d0, p["final.weight"], p["final.bias"], 1, 1)
    if taps is not None:
        taps["dec1"], taps["final"] = d0, logits
    out = torch.sigmoid(logits)
    return out * (1 - mask) + x * mask


def discriminator_forward(p: Params, img: Tensor, training: bool = True) -> Tensor:
    """Discriminator.forward -- discriminator.py:17-26."""
    h = img
    for idx, _ci, _co, _k, s, pad, bn, leaky in D_LAYERS:
        h = F.conv2d(h, p[f"model.{idx}.weight"], p[f"model.{idx}.bias"], s, pad)
        if bn is not None:
            h = batch_norm(h, p, f"model.{bn}", training)
        if leaky:
            h = F.leaky_relu(h, 0.2)
    return h


def vgg_features(vp: Params, x1: Tensor) -> Tensor:
    """features[:16] applied to the 1-channel image repeated x3, un-normalised (losses.py:79-88)."""
    h = x1.repeat(1, 3, 1, 1)
    for item in VGG_TRUNK:
        if item == "M":
            h = F.max_pool2d(h, 2, 2)
        else:
            idx = item[0]
            h = F.relu(F.conv2d(h, vp[f"{idx}.weight"], vp[f"{idx}.bias"], 1, 1))
    return h


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------
def tv_loss(x: Tensor) -> Tensor:
    """losses.py:118-127 (divides by B twice: count_* already contains B)."""
    b, h, w = x.shape[0], x.shape[2], x.shape[3]
    count_h = x[:, :, 1:, :].numel()
    count_w = x[:, :, :, 1:].numel()
    h_tv = ((x[:, :, 1:, :] - x[:, :, :h - 1, :]) ** 2).sum()
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :w - 1]) ** 2).sum()
    return 2 * (h_tv / count_h + w_tv / count_w) / b


def boundary_band(mask: Tensor) -> Tensor:
    """losses.py:406-408: 3x3 morphological gradient of the mask."""
    dil = F.max_pool2d(mask, 3, 1, 1)
    ero = 1 - F.max_pool2d(1 - mask, 3, 1, 1)
    return torch.clamp(dil - ero, 0.0, 1.0)


def boundary_loss(pred: Tensor, target: Tensor, mask: Tensor, eps: float = 1e-6) -> Tensor:
    """BoundaryAwareLoss.forward -- losses.py:386-428 (empty band -> 0, NaN/Inf -> 0)."""
    band = boundary_band(mask)
    if float(band.sum()) < 1.0:
        return torch.zeros(())
    loss = ((pred - target).abs() * band).sum() / (band.sum() + eps)
    if bool(torch.isnan(loss)) or bool(torch.isinf(loss)):
        return torch.zeros(())
    return loss


def inpainting_loss(vp: Params, pred: Tensor, target: Tensor, mask: Tensor, w_perc: float = 0.1,
                    w_tv: float = 0.1, w_bnd: float = 0.5) -> Tuple[Tensor, Dict[str, Tensor]]:
    """InpaintingLoss.forward -- losses.py:58-116."""
    parts: Dict[str, Tensor] = {}
    parts["l1"] = (pred - target).abs().mean()
    total = parts["l1"]
    if w_perc > 0:
        parts["perc"] = (vgg_features(vp, pred) - vgg_features(vp, target)).abs().mean()
        total = total + w_perc * parts["perc"]
    if w_tv > 0:
        parts["tv"] = tv_loss(pred * (1 - mask))
        total = total + w_tv * parts["tv"]
    if w_bnd > 0:
        parts["boundary"] = boundary_loss(pred, target, mask)
        total = total + w_bnd * parts["boundary"]
    return total, parts


def human_guided_loss(vp: Params, pred: Tensor, target: Tensor, mask: Tensor,
                      human_mask: Optional[Tensor], w_base: float = 0.7, w_human: float = 0.3,
                      w_perc: float = 0.1, w_tv: float = 0.1, w_bnd: float = 0.5) -> Tensor:
    """HumanGuidedLoss.forward -- losses.py:152-204."""
    base, _ = inpainting_loss(vp, pred, target, mask, w_perc, w_tv, w_bnd)
    human = torch.zeros(())
    if human_mask is not None:
        h = (human_mask > 0).float()
        if float(h.sum()) > 0:
            human = (pred * h - target * h).abs().mean()
            if w_bnd > 0:
                human = human + w_bnd * boundary_loss(pred, target, h)
    return w_base * base + w_human * human


def bce_logits(logits: Tensor, target_value: float) -> Tensor:
    """nn.BCEWithLogitsLoss (mean) against a constant target -- train.py:115,203,215-216."""
    return F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, target_value))


# ----------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults: betas .9/.999, eps 1e-8, no wd) -- main_pipeline.py:214-221
# ----------------------------------------------------------------------------
class Adam:
    def __init__(self, params: Params, keys: List[str], lr: float = 2e-4,
                 betas=(0.9, 0.999), eps: float = 1e-8):
        self.p, self.keys, self.lr, self.betas, self.eps = params, keys, lr, betas, eps
        self.m = {k: torch.zeros_like(params[k]) for k in keys}
        self.v = {k: torch.zeros_like(params[k]) for k in keys}
        self.t = 0

    @torch.no_grad()
    def step(self, grads: Dict[str, Tensor]) -> None:
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k in self.keys:
            g = grads.get(k)
            if g is None:
                continue
            self.m[k].lerp_(g, 1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            self.p[k].addcdiv_(self.m[k], denom, value=-self.lr / bc1)


# ----------------------------------------------------------------------------
# the train step -- train.py:177-219
# ----------------------------------------------------------------------------
def _grads(loss: Tensor, p: Params, keys: List[str]) -> Dict[str, Tensor]:
    gs = torch.autograd.grad(loss, [p[k] for k in keys], allow_unused=True)
    return {k: g for k, g in zip(keys, gs) if g is not None}


def g_step_grads(gp: Params, dp: Params, vp: Params, real: Tensor, mask: Tensor,
                 w_perc=0.1, w_tv=0.1, w_bnd=0.5):
    """Generator half of the step (train.py:179-204) up to the gradients.
    D runs in train mode, so its BN running stats move here too."""
    gk = trainable(gp)
    for k in gk:
        gp[k].requires_grad_(True)
    for k in trainable(dp):
        dp[k].requires_grad_(False)
    masked = real * mask
    gen = generator_forward(gp, masked, mask, True)
    g_loss, parts = inpainting_loss(vp, gen, real, mask, w_perc, w_tv, w_bnd)
    g_adv = bce_logits(discriminator_forward(dp, gen, True), 1.0)
    total = g_loss + g_adv
    grads = _grads(total, gp, gk)
    for k in gk:
        gp[k].requires_grad_(False)
    scal = {"g_total": total.detach(), "g_loss": g_loss.detach(), "g_adv": g_adv.detach()}
    scal.update({k: v.detach() for k, v in parts.items()})
    return gen.detach(), grads, scal


def d_step_grads(dp: Params, real: Tensor, gen: Tensor):
    """Discriminator half (train.py:210-218) up to the gradients."""
    dk = trainable(dp)
    for k in dk:
        dp[k].requires_grad_(True)
    real_loss = bce_logits(discriminator_forward(dp, real, True), 1.0)
    fake_loss = bce_logits(discriminator_forward(dp, gen, True), 0.0)
    d_loss = 0.5 * (real_loss + fake_loss)
    grads = _grads(d_loss, dp, dk)
    for k in dk:
        dp[k].requires_grad_(False)
    return grads, {"d_loss": d_loss.detach(), "real_loss": real_loss.detach(),
                   "fake_loss": fake_loss.detach()}


class TrainState:
    """torch.manual_seed(seed) -> G, D, criterion (stand-in VGG) in that order."""

    def __init__(self, seed: int = 0, lr: float = 2e-4, w_perc=0.1, w_tv=0.1, w_bnd=0.5):
        torch.manual_seed(seed)
        self.gp = init_generator()
        self.dp = init_discriminator()
        self.vp = init_vgg_standin()
        self.opt_g = Adam(self.gp, trainable(self.gp), lr)
        self.opt_d = Adam(self.dp, trainable(self.dp), lr)
        self.w = (w_perc, w_tv, w_bnd)

    def to(self, dtype):
        """The same state in another floating-point type (fresh optimisers).  float64 gives the reference arithmetic's
        own fp32-vs-fp64 deviation: the reference cannot run in fp64 itself (pconv.py:35,40 hard-code .float())."""
        for d in (self.gp, self.dp, self.vp):
            for k in d:
                if d[k].dtype.is_floating_point:
                    d[k] = d[k].to(dtype)
        lr = self.opt_g.lr
        self.opt_g = Adam(self.gp, trainable(self.gp), lr)
        self.opt_d = Adam(self.dp, trainable(self.dp), lr)
        return self


def train_step(st: TrainState, real: Tensor, mask: Tensor):
    """One GAN step in the reference's order (train.py:177-219)."""
    gen, gg, gs = g_step_grads(st.gp, st.dp, st.vp, real, mask, *st.w)
    st.opt_g.step(gg)
    dg, ds = d_step_grads(st.dp, real, gen)
    st.opt_d.step(dg)
    gs.update(ds)
    return gen, gs, gg, dg


def dp_train_step(st: TrainState, reals: List[Tensor], masks: List[Tensor]):
    """Data-parallel semantics: N micro-batches from identical weights, mean of the
    gradients, ONE Adam step per optimiser.  BN running stats follow rank 0 (not synchronised)."""
    import copy
    n = len(reals)
    gens, gsum, scal = [], None, []
    bn0 = None
    for r in range(n):
        gp_r = {k: v.clone() for k, v in st.gp.items()} if r else st.gp
        dp_r = {k: v.clone() for k, v in st.dp.items()} if r else st.dp
        gen, gg, gs = g_step_grads(gp_r, dp_r, st.vp, reals[r], masks[r], *st.w)
        gens.append(gen)
        scal.append(gs)
        gsum = gg if gsum is None else {k: gsum[k] + gg[k] for k in gsum}
    st.last_gg = {k: v / n for k, v in gsum.items()}       # the averaged gradients (what the all-reduce delivers)
    st.opt_g.step(st.last_gg)
    dsum = None
    for r in range(n):
        dp_r = {k: v.clone() for k, v in st.dp.items()} if r else st.dp
        dg, ds = d_step_grads(dp_r, reals[r], gens[r])
        scal[r].update(ds)
        dsum = dg if dsum is None else {k: dsum[k] + dg[k] for k in dsum}
    st.last_dg = {k: v / n for k, v in dsum.items()}
    st.opt_d.step(st.last_dg)
    return gens, scal


# ----------------------------------------------------------------------------
# validation pass and logged quality metrics
# ----------------------------------------------------------------------------
@torch.no_grad()
def validation_losses(gp: Params, dp: Params, vp: Params, real: Tensor, mask: Tensor,
                      w_perc=0.1, w_tv=0.1, w_bnd=0.5) -> Tuple[Tensor, Tensor, Tensor]:
    """Validation body -- train.py:283-301: generator in EVAL mode (running statistics), criterion, then D(real) and
    D(gen) with the discriminator left in TRAIN mode (train.py:279 switches only the generator), so D's BatchNorm running
    statistics move during validation too.  Returns (val_g_total, val_d, gen)."""
    gen = generator_forward(gp, real * mask, mask, False)
    g_total, _ = inpainting_loss(vp, gen, real, mask, w_perc, w_tv, w_bnd)
    d_real = bce_logits(discriminator_forward(dp, real, True), 1.0)
    d_fake = bce_logits(discriminator_forward(dp, gen, True), 0.0)
    return g_total, 0.5 * (d_real + d_fake), gen


def boundary_quality(pred: Tensor, target: Tensor, mask: Tensor) -> Dict[str, float]:
    """calculate_boundary_quality -- mvp_gan/src/evaluation/metrics.py:79-133.  The MSE is a mean over ALL elements of
    ((pred-target)*band)^2 (:101), not over the band; `boundary_width` is unused there."""
    band = boundary_band(mask)                                                  # :88-90
    if float(band.sum()) < 1e-6:                                                # :93-98
        return {"boundary_mse": 0.0, "boundary_psnr": 0.0, "boundary_gradient_diff": 0.0}
    mse = (((pred - target) * band) ** 2).mean()                                # :101
    psnr = 10 * torch.log10(1.0 / (mse + 1e-6))                                 # :104-106
    pd = (pred[:, :, 1:, :] - pred[:, :, :-1, :]).abs().mean() + (pred[:, :, :, 1:] - pred[:, :, :, :-1]).abs().mean()
    td = (target[:, :, 1:, :] - target[:, :, :-1, :]).abs().mean() + (target[:, :, :, 1:] - target[:, :, :, :-1]).abs().mean()
    return {"boundary_mse": float(mse), "boundary_psnr": float(psnr), "boundary_gradient_diff": float((pd - td).abs())}


def psnr(pred: Tensor, target: Tensor) -> float:
    """utils/experiment_tracking.py:196-206 (= MaskEvaluator._calculate_psnr, evaluation/metrics.py:47-54)."""
    mse = F.mse_loss(pred, target)
    if float(mse) == 0:
        return float("inf")
    return float(20 * torch.log10(1.0 / torch.sqrt(mse)))


def ssim(pred: Tensor, target: Tensor, window: int = 11) -> float:
    """utils/experiment_tracking.py:209-231: box-window SSIM, avg_pool2d with zero padding (divisor window^2)."""
    c1, c2, pad = 0.01 ** 2, 0.03 ** 2, window // 2
    mu1, mu2 = F.avg_pool2d(pred, window, 1, pad), F.avg_pool2d(target, window, 1, pad)
    s1 = F.avg_pool2d(pred * pred, window, 1, pad) - mu1 * mu1
    s2 = F.avg_pool2d(target * target, window, 1, pad) - mu2 * mu2
    s12 = F.avg_pool2d(pred * target, window, 1, pad) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s1 + s2 + c2))
    return float(m.mean())


def l1_l2(pred: Tensor, target: Tensor) -> Tuple[float, float]:
    """utils/experiment_tracking.py:176-192."""
    return float((pred - target).abs().mean()), float(F.mse_loss(pred, target).sqrt())


# ----------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------
def synth_batch(batch: int, size: int, seed: int) -> Tuple[Tensor, Tensor]:
    """DSM tile in [0,1] quantised to k/255 + disc-hole mask (1 = valid).  CPU generator so the
    same tensors are produced on every machine."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand(batch, 1, 8, 8, generator=g)
    dsm = F.interpolate(coarse, size=(size, size), mode="bilinear", align_corners=False)
    dsm = dsm + 0.05 * torch.rand(batch, 1, size, size, generator=g)
    lo = dsm.amin(dim=(2, 3), keepdim=True)
    hi = dsm.amax(dim=(2, 3), keepdim=True)
    dsm = torch.round((dsm - lo) / (hi - lo) * 255.0) / 255.0
    mask = torch.ones(batch, 1, size, size)
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    for b in range(batch):
        k = int(torch.randint(3, 13, (1,), generator=g))
        for _ in range(k):
            r = int(torch.randint(10, 51, (1,), generator=g)) * size / 500.0
            cy = int(torch.randint(0, size, (1,), generator=g))
            cx = int(torch.randint(0, size, (1,), generator=g))
            mask[b, 0][((yy - cy) ** 2 + (xx - cx) ** 2).float() <= r * r] = 0.0
    return dsm.contiguous(), mask.contiguous()
