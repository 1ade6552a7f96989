"""Explicit forward/backward schedules of the hot path on the HIP kernels (no autograd inside).

Each engine function takes the module's tensors as a dict keyed like the reference state-dict
(`enc1.input_conv.weight`, `model.3.running_mean`, ...) and returns a context object that the
matching backward consumes.  Activations are [B][H][W][C]; 1-channel images/masks are [B][H][W].

Reference arithmetic reproduced (paths relative to /root/reference):
  generator_*      mvp_gan/src/models/generator.py:31-84 + pconv.py:25-50
  discriminator_*  mvp_gan/src/models/discriminator.py:17-26
  vgg_*            torchvision vgg16.features[:16] as used by mvp_gan/src/utils/losses.py:31-34,79-90
"""
import os
from types import SimpleNamespace as NS

import torch

from . import ops as O

# (name, cin, cout, k, stride, pad) -- generator.py:13-28
G_ENC = [("enc1", 1, 64, 7, 2, 3), ("enc2", 64, 128, 5, 2, 2), ("enc3", 128, 256, 5, 2, 2),
         ("enc4", 256, 512, 3, 2, 1), ("enc5", 512, 512, 3, 2, 1), ("enc6", 512, 512, 3, 2, 1),
         ("enc7", 512, 512, 3, 2, 1)]
G_DEC = [("dec7", 1024, 512, 3, 1, 1), ("dec6", 1024, 512, 3, 1, 1), ("dec5", 1024, 512, 3, 1, 1),
         ("dec4", 768, 256, 3, 1, 1), ("dec3", 384, 128, 3, 1, 1), ("dec2", 192, 64, 3, 1, 1),
         ("dec1", 64, 64, 3, 1, 1)]
# (conv index, bn index or None, cin, cout, k, stride, pad, leaky) -- discriminator.py:17-23
D_LAYERS = [(0, None, None, 64, 4, 2, 1, True), (2, 3, 64, 128, 4, 2, 1, True), (5, 6, 128, 256, 4, 2, 1, True),
            (8, 9, 256, 512, 4, 2, 1, True), (11, None, 512, 1, 4, 1, 1, False)]
# features[:16]: conv indices, 'M' = 2x2 max-pool; every conv is followed by ReLU
VGG_TRUNK = [0, 2, "M", 5, 7, "M", 10, 12, 14]


# --------------------------------------------------------------------------------------------------
# side stream for weight gradients
# --------------------------------------------------------------------------------------------------
# A weight gradient is a leaf of the backward graph: nothing downstream waits for it until the optimiser, so it can go to a
# second HIP stream and run underneath the backward chain (BN-backward reductions, activation gates, dgrad).  Measured
# (round 1, 1x MI355X): +0.9 % only -- the Winograd wgrad workgroups own a whole CU each (141 KB LDS, 2 x 256 registers
# per SIMD) for ~300 us, so the chain's small kernels queue behind them (rocprofv3: a 21 us BN reduction stretched to
# 380 us) and the overlap is paid back on the critical path.  Kept as an opt-in (TG_SIDE_STREAM=1) until the wgrad grid
# can be confined to a CU subset.  Only used with persistent gradient buffers (outputs never owned by the side stream's
# allocator pool); inputs are pinned with record_stream so the caching allocator does not recycle them.
# The frozen VGG16 trunk (perceptual loss only) and Winograd F(4x4,3x3) (csrc/wino44.inc: 1.78x fewer multiplies than F(2x2,3x3),
# 6-30x its rounding error per layer -- 1e-6 ... 1e-5 of the tensor's largest value).
#   * FORWARD: F(2x2,3x3).  The perceptual term is mean |VGG(pred) - VGG(target)| (losses.py:79-90) and its gradient is
#     sign(fp - ft) pushed back through the trunk; pred equals target outside the holes, so fp - ft is SMALL against the features
#     over most of the affected region and F(4x4)'s forward error flips those signs wholesale: measured (round 4,
#     tools/wino4_admission_probe.py, profiles/r04_wino4_probe.json) 21-33 % rms error of d perceptual / d pred against fp64,
#     for default-initialised and for trained-like weights alike, where F(2x2,3x3) sits at the CPU fp32 evaluation's own error
#     (0.2-5 %).  Rounds 1-3 had admitted the forward on the loss VALUE and on the train-step fixtures, where at initialisation
#     the perceptual part of dL/dgen is negligible.  TG_VGG_WINO4_FWD=1 opts back in (throughput experiments only).
#   * DGRAD: F(4x4,3x3) stays (TG_VGG_WINO4=0 turns it off): the backward is a LINEAR map of the sign pattern, its error enters
#     once, at 1e-5 of max -- forward F(2x2) + dgrad F(4x4) gives the same gradient error as F(2x2) throughout, to 6 digits.
#     Per layer only where a side has >= 128 channels: on the 64 -> 64 layer (8 K steps per work item) the heavier output
#     transform eats the gain (measured: dgrad 0.37 -> 0.40 ms).
VGG_WINO4 = os.environ.get("TG_VGG_WINO4", "1") != "0"
VGG_WINO4_FWD = os.environ.get("TG_VGG_WINO4_FWD", "0") == "1"
VGG_WINO4_MINCH = int(os.environ.get("TG_VGG_WINO4_MINCH", "128"))
POOL_CODE = os.environ.get("TG_NO_POOL_CODE") is None       # pooled convs of the trunk: pooled tensor + pool code, no full-resolution output


def _vgg_wino4(w, B, H, W, cout, mode=None, default=True):
    """F(4x4,3x3) for this launch?  mode None: `default` and only when the 16 x 32-pixel x 64-channel work items of that kernel
    fill the chip (small batches: the F(2x2,3x3) kernel's 16 x 16 items do better); True: wherever the geometry allows (tests,
    probes); False: never."""
    if mode is False or (mode is None and not default):
        return False
    wide = max(w.shape[0], w.shape[1]) >= VGG_WINO4_MINCH
    if mode is True:
        return wide
    items = B * ((H + 15) // 16) * ((W + 31) // 32) * (cout // 64)
    return wide and items >= 256
_side = {}
SIDE_WGRAD = os.environ.get("TG_SIDE_STREAM") == "1"
# Diagnostics (tests/test_hip_backward_chain.py, tools/backward_chain.py): PROBE(kind, name, tensor) is called with every
# generator layer's output activation (kind "fwd") and with the gradient entering every layer's backward ("bwd", BEFORE the
# layer consumes it in place).  None in production: one attribute test per layer.
PROBE = None


def _side_stream():
    dev = torch.cuda.current_device()
    st = _side.get(dev)
    if st is None:
        st = _side[dev] = torch.cuda.Stream(device=dev)
    return st


def _wgrad(gbuf, inputs, fn):
    """fn() launches one weight gradient.  With gradient buffers it runs on the side stream, ordered after everything
    enqueued so far on the current stream; join_side() must be called before the gradients are consumed."""
    if gbuf is None or not SIDE_WGRAD:
        return fn()
    main, side = torch.cuda.current_stream(), _side_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        out = fn()
    for t in inputs:
        if t is not None:
            t.record_stream(side)
    return out


def join_side():
    if SIDE_WGRAD and torch.cuda.current_device() in _side:
        torch.cuda.current_stream().wait_stream(_side_stream())


# --------------------------------------------------------------------------------------------------
# partial-conv layer (conv -> *ratio -> BN -> ReLU)
# --------------------------------------------------------------------------------------------------
def _pconv_fwd(P, name, k, s, p, x, in_mask, ratio, training, premasked=False, defer_act=False):
    """premasked: x already holds input*mask (the decoder concat is written that way), so the conv and its
    wgrad skip the mask prologue; dgrad still applies the mask.
    defer_act: ReLU(BN(y)) is NOT written (returned activation = None): the layer's one consumer applies it while loading y
    (dec1 -> `final`: O.conv_fwd_bnin / conv_wgrad(in_bn=...), with _bn_in(P, ctx))."""
    O.tag(f"{name}.fwd")
    y = O.conv_fwd(x, P[f"{name}.input_conv.weight"], P[f"{name}.input_conv.bias"].detach(), k, s, p,
                   in_mask=None if premasked else in_mask, ratio=ratio)
    if training:
        mean, rstd, a = O.bn_fwd(y, P[f"{name}.bn.weight"], P[f"{name}.bn.bias"], O.ACT_RELU, 0.0, P[f"{name}.bn.running_mean"],
                                 P[f"{name}.bn.running_var"], P[f"{name}.bn.num_batches_tracked"], apply=not defer_act)
    else:
        mean, rstd = O.bn_eval_stats(P[f"{name}.bn.running_mean"], P[f"{name}.bn.running_var"])
        a = None if defer_act else O.bn_act_fwd(y, mean, rstd, P[f"{name}.bn.weight"], P[f"{name}.bn.bias"], O.ACT_RELU)
    if PROBE is not None:
        PROBE("fwd", name, a if a is not None else O.bn_act_fwd(y, mean, rstd, P[f"{name}.bn.weight"], P[f"{name}.bn.bias"], O.ACT_RELU))
    return a, NS(name=name, k=k, s=s, p=p, x=x, in_mask=in_mask, ratio=ratio, y=y, mean=mean, rstd=rstd,
                 premasked=premasked)


def _notify(on_ready, grads, keys):
    """Data parallel: tell the bucket launcher (tg_hip.dist.BucketLauncher.ready) that these gradients are enqueued.
    Not with the side-stream option -- the collective is ordered behind the CURRENT stream only."""
    if on_ready is not None and not SIDE_WGRAD:
        for k in keys:
            on_ready(k, grads[k])


def _pconv_bwd(P, c, da, grads, dx_out=None, want_dx=True, gbuf=None, on_ready=None, conv1=None):
    """da: grad w.r.t. the layer's ReLU output (consumed in place).  Returns dx (or None).
    gbuf: {state-dict key: preallocated gradient tensor} (tg_hip.gradbuf) -- written in place when given.
    conv1 = (dz, w): da is None and stands for conv_dgrad(dz, w) of the C -> 1 channel conv above this layer (`final` over dec1),
    recomputed inside the BatchNorm backward instead of being materialised (O.bn_act_bwd_conv1)."""
    name = c.name
    if PROBE is not None:
        PROBE("bwd", name, da)
    outs = (gbuf[f"{name}.bn.weight"], gbuf[f"{name}.bn.bias"], gbuf[f"{name}.input_conv.bias"]) if gbuf is not None else None
    if conv1 is not None:
        dyr, dgamma, dbeta, db = O.bn_act_bwd_conv1(conv1[0], conv1[1], c.y, c.mean, c.rstd, P[f"{name}.bn.weight"], P[f"{name}.bn.bias"],
                                                    O.ACT_RELU, ratio=c.ratio, outs=outs)
    else:
        dyr, dgamma, dbeta, db = O.bn_act_bwd(da, c.y, c.mean, c.rstd, P[f"{name}.bn.weight"], P[f"{name}.bn.bias"], O.ACT_RELU,
                                              ratio=c.ratio, outs=outs)
    # dgrad first: the weight gradient is enqueued behind it on the side stream, so it starts when the dgrad has
    # finished and runs underneath the NEXT layer's BatchNorm-backward reductions (small grids) and its dgrad
    O.tag(f"{name}.dgrad")
    dx = O.conv_dgrad(dyr, P[f"{name}.input_conv.weight"], tuple(c.x.shape), c.k, c.s, c.p, in_mask=c.in_mask, out=dx_out) \
        if want_dx else None
    wmask = None if getattr(c, "premasked", False) else c.in_mask
    O.tag(f"{name}.wgrad")
    dw, _ = _wgrad(gbuf, (c.x, dyr, wmask), lambda: O.conv_wgrad(
        c.x, dyr, P[f"{name}.input_conv.weight"], c.k, c.s, c.p, in_mask=wmask, want_bias=False,
        dw_out=gbuf[f"{name}.input_conv.weight"] if gbuf is not None else None))
    grads[f"{name}.input_conv.weight"], grads[f"{name}.input_conv.bias"] = dw, db
    grads[f"{name}.bn.weight"], grads[f"{name}.bn.bias"] = dgamma, dbeta
    # reverse registration order = the order of the flat gradient buffer (bn.bias, bn.weight, conv bias, conv weight)
    _notify(on_ready, grads, (f"{name}.bn.bias", f"{name}.bn.weight", f"{name}.input_conv.bias", f"{name}.input_conv.weight"))
    return dx


# --------------------------------------------------------------------------------------------------
# generator
# --------------------------------------------------------------------------------------------------
def generator_forward(P, x, mask, training=True, checkpoint=False, out=None):
    """x (already masked image) and mask: [B][H][W].  Returns (out [B][H][W], ctx).  `out`: preallocated destination of the
    generated batch (train_step: the first half of the stacked [gen; real] buffer).
    checkpoint=True (activation checkpointing, BASELINE config 5): only the pre-BatchNorm conv outputs, the batch
    statistics and the (1-channel) masks are kept; the post-activation tensors and the decoder concat tensors --
    more than half of the footprint -- are dropped and recomputed in backward (one BN+ReLU pass and one
    upsample/concat pass per layer, ~3 % of a step)."""
    B, H, W = x.shape
    # mask pyramid first: it depends on the input mask only (pconv.py:33-40, generator.py:51-54,68,74) -- 14 mask updates
    # and 7 up-merges from ONE launch
    m, er, dmasks, dr = O.mask_pyramid(mask, [(k, s, p) for (_n, _ci, _co, k, s, p) in G_ENC],
                                       [(k, s, p) for (_n, _ci, _co, k, s, p) in G_DEC])

    enc_ctx, e = [], [x.reshape(B, H, W, 1)]
    for i, (name, _ci, _co, k, s, p) in enumerate(G_ENC):
        a, c = _pconv_fwd(P, name, k, s, p, e[-1], m[i], er[i + 1], training)
        e.append(a)
        enc_ctx.append(c)
    d, dec_ctx = e[7], []
    fshape = (B, H, W, G_DEC[6][2])
    bnin_final = BNIN_FINAL and O.conv_bnin_supported(fshape, 1, 3, 1, 1) and O.conv_bnin_supported(fshape, 1, 3, 1, 1, wgrad=True)
    up_c = None                                                      # context of the layer whose ReLU(BN(y)) was deferred to this upcat
    for i, (name, _ci, _co, k, s, p) in enumerate(G_DEC):
        skip = e[6 - i] if i < 6 else None
        Hs, Ws = (skip.shape[1], skip.shape[2]) if skip is not None else (H, W)
        if up_c is not None:
            up_shape = tuple(up_c.y.shape)
            cat = O.upcat_fwd(up_c.y, skip, Hs, Ws, out_mask=dmasks[i], up_bn=_bn_in(P, up_c))
        else:
            up_shape = tuple(d.shape)
            cat = O.upcat_fwd(d, skip, Hs, Ws, out_mask=dmasks[i])   # = merged_feature * merged_mask
        if i == 6:
            defer = bnin_final
        else:
            # this layer's activation has one reader, the next level's upsample: deferred to it where that kernel can (exact x2
            # levels) and the BatchNorm is a two-launch one anyway (the small maps' one-launch form already applies it)
            nskip = e[5 - i] if i < 5 else None
            nH, nW = (nskip.shape[1], nskip.shape[2]) if nskip is not None else (H, W)
            defer = BNIN_UPCAT and Hs * Ws * B > O.BN_SMALL_ROWS and \
                O.upcat_bn_supported((B, Hs, Ws, _co), None if nskip is None else tuple(nskip.shape), nH, nW)
        d, c = _pconv_fwd(P, name, k, s, p, cat, dmasks[i], dr[i], training, premasked=True, defer_act=defer)
        up_c = c if (defer and i < 6) else None
        c.up_shape, c.x_shape, c.skip_hw = up_shape, tuple(cat.shape), (Hs, Ws)
        if checkpoint:
            c.x = None                                               # concat tensor: rebuilt in backward
        dec_ctx.append(c)
    O.tag("final.fwd")
    if bnin_final:                                                                    # generator.py:29,56
        logits = O.conv_fwd_bnin(dec_ctx[6].y, _bn_in(P, dec_ctx[6]), P["final.weight"], P["final.bias"].detach(), 3, 1, 1)
    else:
        logits = O.conv_fwd(d, P["final.weight"], P["final.bias"].detach(), 3, 1, 1)
    out = O.sigmoid_composite_fwd(logits.reshape(B, H, W), x, mask, out=out)            # generator.py:57-62
    if PROBE is not None:
        PROBE("fwd", "final", logits)
        PROBE("fwd", "gen", out)
    if checkpoint:
        for i in range(1, 7):
            enc_ctx[i].x_shape = tuple(enc_ctx[i].x.shape)
            enc_ctx[i].x = None                                      # = ReLU(BN(y)) of the previous encoder layer
        d = None
    return out, NS(enc=enc_ctx, dec=dec_ctx, d0=d, logits=logits, mask=mask, shape=(B, H, W), checkpoint=checkpoint,
                   bnin_final=bnin_final)


def _act_of(P, c):
    """Recompute a layer's ReLU(BN(y)) from its kept pre-BN output and batch statistics."""
    return O.bn_act_fwd(c.y, c.mean, c.rstd, P[f"{c.name}.bn.weight"], P[f"{c.name}.bn.bias"], O.ACT_RELU)


def _bn_in(P, c):
    """The BatchNorm + ReLU of layer context c as an `in_bn` tuple (applied on load by the consumer of c.y)."""
    return (c.mean, c.rstd, P[f"{c.name}.bn.weight"], P[f"{c.name}.bn.bias"], O.ACT_RELU, 0.0)


# dec1's ReLU(BN(.)) output has ONE consumer per pass, the 64 -> 1 `final` conv (forward: the conv; backward: its weight gradient),
# an HBM-bound kernel that stages its source through LDS: the affine map + ReLU are applied there and the widest activation of the
# network (B x H x W x 64) is neither written nor read back (TG_NO_BNIN=1: the two-pass form; same bits either way)
BNIN_FINAL = os.environ.get("TG_NO_BNIN") is None
BN_CONV1 = os.environ.get("TG_NO_BN_CONV1") is None        # dec1's BatchNorm backward recomputes `final`'s input gradient (see generator_backward)
# dec2 ... dec5 the same way into the next level's upsample + concat (TG_NO_BNIN_UPCAT=1 / TG_NO_BNIN=1: off)
BNIN_UPCAT = os.environ.get("TG_NO_BNIN") is None and os.environ.get("TG_NO_BNIN_UPCAT") is None


def generator_backward(P, ctx, dout, want_dx=False, gbuf=None, on_ready=None):
    """dout: [B][H][W].  Returns (grads dict keyed like the state-dict, dx or None).  With `gbuf` (persistent gradient
    views, tg_hip.gradbuf) the gradients are written there and the returned dict holds those same tensors."""
    B, H, W = ctx.shape
    grads = {}
    ckpt = getattr(ctx, "checkpoint", False)
    dz, dx_comp = O.sigmoid_composite_bwd(dout, ctx.logits.reshape(B, H, W), ctx.mask, want_dx)
    dz = dz.reshape(B, H, W, 1)
    if PROBE is not None:
        PROBE("bwd", "gen", dout)
        PROBE("bwd", "final", dz)
    bnin = getattr(ctx, "bnin_final", False)
    # (BN-on-load: `final`'s input is dec1's PRE-BatchNorm output + its statistics -- nothing to recompute under checkpointing)
    d0 = ctx.dec[6].y if bnin else (_act_of(P, ctx.dec[6]) if ckpt else ctx.d0)
    # `final`'s input gradient costs nine FMAs per element from the 1-channel dz: dec1's BatchNorm backward recomputes it in both
    # of its passes instead of reading a written copy twice (TG_NO_BN_CONV1=1 / a PROBE: the materialised form)
    fuse1 = BN_CONV1 and PROBE is None and O.bn_bwd_conv1_supported(tuple(d0.shape))
    if fuse1:
        da = None
    else:
        O.tag("final.dgrad")
        da = O.conv_dgrad(dz, P["final.weight"], tuple(d0.shape), 3, 1, 1)
    O.tag("final.wgrad")
    grads["final.weight"], grads["final.bias"] = _wgrad(gbuf, (d0, dz), lambda: O.conv_wgrad(
        d0, dz, P["final.weight"], 3, 1, 1, dw_out=gbuf["final.weight"] if gbuf is not None else None,
        db_out=gbuf["final.bias"] if gbuf is not None else None, in_bn=_bn_in(P, ctx.dec[6]) if bnin else None))
    _notify(on_ready, grads, ("final.bias", "final.weight"))
    del d0
    dskips = {}
    for i in range(6, -1, -1):                     # dec1 ... dec7
        c = ctx.dec[i]
        if ckpt:                                   # rebuild this layer's (pre-masked) concat input
            below = ctx.dec[i - 1] if i > 0 else ctx.enc[6]
            skip = _act_of(P, ctx.enc[5 - i]) if i < 6 else None
            if BNIN_UPCAT and O.upcat_bn_supported(tuple(below.y.shape), None if skip is None else tuple(skip.shape), *c.skip_hw):
                c.x = O.upcat_fwd(below.y, skip, c.skip_hw[0], c.skip_hw[1], out_mask=c.in_mask, up_bn=_bn_in(P, below))
            else:
                up_src = _act_of(P, below)
                c.x = O.upcat_fwd(up_src, skip, c.skip_hw[0], c.skip_hw[1], out_mask=c.in_mask)
                del up_src
            del skip
        dcat = _pconv_bwd(P, c, da, grads, gbuf=gbuf, on_ready=on_ready, conv1=(dz, P["final.weight"]) if (i == 6 and fuse1) else None)
        if ckpt:
            c.x = None
        _b, h, w, Cu = c.up_shape
        da, dskip = O.upcat_bwd(dcat, h, w, Cu)
        if i < 6:
            dskips[6 - i] = dskip                  # gradient reaching encoder output e[6-i] through the skip
    # da is now the gradient of e7 (through dec7's upsample path only)
    dx = None
    for i in range(6, -1, -1):                     # enc7 ... enc1
        c = ctx.enc[i]
        if ckpt and i > 0:
            c.x = _act_of(P, ctx.enc[i - 1])
        if i > 0:
            # gradient of e[i] = skip part (already there) + this layer's dgrad, accumulated in place
            da = _pconv_bwd(P, c, da, grads, dx_out=dskips[i], gbuf=gbuf, on_ready=on_ready)
            if ckpt:
                c.x = None
        else:
            dx = _pconv_bwd(P, c, da, grads, want_dx=want_dx, gbuf=gbuf, on_ready=on_ready)
    if want_dx:
        dx = O.axpby_(dx_comp, 1.0, 1.0, dx.reshape(B, H, W))
    join_side()
    return grads, dx


# --------------------------------------------------------------------------------------------------
# discriminator
# --------------------------------------------------------------------------------------------------
def discriminator_forward(P, img, training=True, groups=1, update_running=True):
    """img [B][H][W] (1 channel) or [B][H][W][C].  Returns (logits [B][h][w][1], ctx).

    groups > 1: the batch is `groups` independent passes stacked along B (train_step stacks D(fake) and D(real),
    which share weights: train.py:202,211): convolutions run once over the whole stack, BatchNorm statistics are taken per
    group exactly as the separate passes would.  update_running=False leaves the running statistics alone (the caller
    replays the updates in the reference's order with discriminator_replay_running_stats)."""
    h = img if img.dim() == 4 else img.reshape(*img.shape, 1)
    Bg = h.shape[0] // groups
    assert Bg * groups == h.shape[0]
    layers = []
    for (ci, bi, _cin, _cout, k, s, p, leaky) in D_LAYERS:
        w, b = P[f"model.{ci}.weight"], P[f"model.{ci}.bias"].detach()
        O.tag(f"d{ci}.fwd")
        if bi is None:
            act = O.ACT_LEAKY if leaky else O.ACT_NONE
            a = O.conv_fwd(h, w, b, k, s, p, act=act, slope=0.2)
            layers.append(NS(ci=ci, bi=None, k=k, s=s, p=p, x=h, a=a, act=act))
        else:
            y = O.conv_fwd(h, w, b, k, s, p)
            if groups > 1 and training and y.shape[-1] % 4 == 0:
                # statistics per pass, ONE set of launches for all passes (tg_bn_fwd_grouped); running statistics by replay
                mean2, rstd2, a = O.bn_fwd_grouped(y, groups, P[f"model.{bi}.weight"], P[f"model.{bi}.bias"], O.ACT_LEAKY, 0.2)
                if update_running:
                    O.bn_running_update_multi(mean2, rstd2, y.numel() // y.shape[-1] // groups, list(range(groups)),
                                              P[f"model.{bi}.running_mean"], P[f"model.{bi}.running_var"],
                                              P[f"model.{bi}.num_batches_tracked"])
                means, rstds = [mean2[gi] for gi in range(groups)], [rstd2[gi] for gi in range(groups)]
                layers.append(NS(ci=ci, bi=bi, k=k, s=s, p=p, x=h, y=y, mean=means[0], rstd=rstds[0], means=means, rstds=rstds,
                                 mean2=mean2, rstd2=rstd2, a=a))
                h = a
                continue
            a = torch.empty_like(y)
            means, rstds = [], []
            for gi in range(groups):
                yg, ag = y[gi * Bg:(gi + 1) * Bg], a[gi * Bg:(gi + 1) * Bg]
                if training:
                    run = (P[f"model.{bi}.running_mean"], P[f"model.{bi}.running_var"], P[f"model.{bi}.num_batches_tracked"]) \
                        if update_running else (None, None, None)
                    mean, rstd = O.bn_stats(yg, *run)
                else:
                    mean, rstd = O.bn_eval_stats(P[f"model.{bi}.running_mean"], P[f"model.{bi}.running_var"])
                O.bn_act_fwd(yg, mean, rstd, P[f"model.{bi}.weight"], P[f"model.{bi}.bias"], O.ACT_LEAKY, 0.2, out=ag)
                means.append(mean), rstds.append(rstd)
            layers.append(NS(ci=ci, bi=bi, k=k, s=s, p=p, x=h, y=y, mean=means[0], rstd=rstds[0], means=means, rstds=rstds, a=a))
        h = a
    return h, NS(layers=layers, groups=groups)


def discriminator_group(ctx, gi):
    """The context of pass `gi` of a grouped forward, as views (what a separate forward of that pass would have kept)."""
    G = ctx.groups
    out = []
    for c in ctx.layers:
        Bg = c.x.shape[0] // G
        sl = slice(gi * Bg, (gi + 1) * Bg)
        d = dict(vars(c))
        d["x"], d["a"] = c.x[sl], c.a[sl]
        if c.bi is not None:
            d["y"], d["mean"], d["rstd"] = c.y[sl], c.means[gi], c.rstds[gi]
            d["means"], d["rstds"] = [c.means[gi]], [c.rstds[gi]]
        out.append(NS(**d))
    return NS(layers=out, groups=1)


def discriminator_backward(P, ctx, dlogits, want_wgrad=True, want_dimg=False, gbuf=None, on_ready=None):
    """Returns (grads dict, dimg [B][H][W][C] or None).  want_wgrad=False skips the parameter
    gradients the reference computes and then discards in the generator step (train.py:204,210)."""
    grads, da, gated = {}, dlogits, False
    for li in range(len(ctx.layers) - 1, -1, -1):
        c = ctx.layers[li]
        w = P[f"model.{c.ci}.weight"]
        db = None
        if c.bi is None:
            # the activation backward of a BN-less block is fused into the dgrad epilogue of the layer above it
            dy = da if (gated or c.act == O.ACT_NONE) else O.act_bwd(da, c.a, c.act, 0.2)
        else:
            outs = (gbuf[f"model.{c.bi}.weight"], gbuf[f"model.{c.bi}.bias"], gbuf[f"model.{c.ci}.bias"]) \
                if (gbuf is not None and want_wgrad) else None
            G_ = getattr(ctx, "groups", 1)
            if G_ > 1 and hasattr(c, "mean2"):
                dy, dgamma, dbeta, db = O.bn_act_bwd_grouped(da, c.y, G_, c.mean2, c.rstd2, P[f"model.{c.bi}.weight"],
                                                             P[f"model.{c.bi}.bias"], O.ACT_LEAKY, 0.2, want_dbias=want_wgrad, outs=outs)
            elif G_ == 1:
                dy, dgamma, dbeta, db = O.bn_act_bwd(da, c.y, c.mean, c.rstd, P[f"model.{c.bi}.weight"], P[f"model.{c.bi}.bias"],
                                                     O.ACT_LEAKY, 0.2, want_dbias=want_wgrad, outs=outs)
            else:
                # BatchNorm backward per pass (its batch sums belong to one pass); the parameter gradients add up
                Bg = da.shape[0] // G_
                dy = da
                for gi in range(G_):
                    sl = slice(gi * Bg, (gi + 1) * Bg)
                    _d, g1, b1, db1 = O.bn_act_bwd(da[sl], c.y[sl], c.means[gi], c.rstds[gi], P[f"model.{c.bi}.weight"],
                                                   P[f"model.{c.bi}.bias"], O.ACT_LEAKY, 0.2, want_dbias=want_wgrad,
                                                   outs=outs if gi == 0 else None)
                    if gi == 0:
                        dgamma, dbeta, db = g1, b1, db1
                    elif want_wgrad:
                        O.axpby_(g1, 1.0, 1.0, dgamma), O.axpby_(b1, 1.0, 1.0, dbeta), O.axpby_(db1, 1.0, 1.0, db)
            if want_wgrad:
                grads[f"model.{c.bi}.weight"], grads[f"model.{c.bi}.bias"] = dgamma, dbeta
        gated = False
        O.tag(f"d{c.ci}.dgrad")
        if li > 0 or want_dimg:
            below = ctx.layers[li - 1] if li > 0 else None
            if below is not None and below.bi is None and below.act != O.ACT_NONE:
                da = O.conv_dgrad(dy, w, tuple(c.x.shape), c.k, c.s, c.p, gate=below.a, gate_act=below.act, gate_slope=0.2)
                gated = True
            else:
                da = O.conv_dgrad(dy, w, tuple(c.x.shape), c.k, c.s, c.p)
        else:
            da = None
        O.tag(f"d{c.ci}.wgrad")
        if want_wgrad:       # behind the dgrad, see _pconv_bwd
            dw, db2 = _wgrad(gbuf, (c.x, dy), lambda c=c, dy=dy, w=w, db=db: O.conv_wgrad(
                c.x, dy, w, c.k, c.s, c.p, want_bias=db is None,
                dw_out=gbuf[f"model.{c.ci}.weight"] if gbuf is not None else None,
                db_out=gbuf[f"model.{c.ci}.bias"] if (gbuf is not None and db is None) else None))
            grads[f"model.{c.ci}.weight"], grads[f"model.{c.ci}.bias"] = dw, (db if db is not None else db2)
            if c.bi is not None:
                _notify(on_ready, grads, (f"model.{c.bi}.bias", f"model.{c.bi}.weight"))
            _notify(on_ready, grads, (f"model.{c.ci}.bias", f"model.{c.ci}.weight"))
    join_side()
    return grads, da


def discriminator_replay_running_stats(P, ctx, order=(0,)):
    """Apply the BN running-stat updates of forward passes whose batch statistics `ctx` holds, in `order` (group indices):
    D(gen.detach()) after D(gen) repeats a pass (train.py:202,212), and a grouped forward leaves all updates to this
    function so that they happen in the reference's order."""
    G_ = getattr(ctx, "groups", 1)
    for c in ctx.layers:
        if c.bi is not None:
            rows = c.y.numel() // c.y.shape[-1] // G_
            if hasattr(c, "mean2"):
                O.bn_running_update_multi(c.mean2, c.rstd2, rows, list(order), P[f"model.{c.bi}.running_mean"],
                                          P[f"model.{c.bi}.running_var"], P[f"model.{c.bi}.num_batches_tracked"])
                continue
            for gi in order:
                mean, rstd = (c.means[gi], c.rstds[gi]) if hasattr(c, "means") else (c.mean, c.rstd)
                O.bn_running_update(mean, rstd, rows, P[f"model.{c.bi}.running_mean"], P[f"model.{c.bi}.running_var"],
                                    P[f"model.{c.bi}.num_batches_tracked"])


# --------------------------------------------------------------------------------------------------
# frozen VGG16 trunk (features[:16]) on a 1-channel image repeated x3 (losses.py:79-90)
# --------------------------------------------------------------------------------------------------
def vgg_forward(V, img, keep=True, wino4=None):
    """V: {'0.weight','0.bias',...,'0.folded'}; img [B][H][W].  Returns (features, ctx).
    wino4 (forward kernels): None = F(2x2,3x3) unless TG_VGG_WINO4_FWD=1 (see the note at the top of this file); True / False
    force F(4x4,3x3) on (wherever the geometry allows) / off."""
    h = img.reshape(*img.shape, 1)
    steps = []
    pooled = code = None
    for i, item in enumerate(VGG_TRUNK):
        if item == "M":
            # (the conv below has written the pooled tensor with its own output where the sizes are even)
            o = pooled if pooled is not None else O.maxpool2_fwd(h)
            if keep:
                steps.append(NS(kind="M", x=h, code=code))
            pooled = code = None
        else:
            w = V["0.folded"] if item == 0 else V[f"{item}.weight"]
            O.tag(f"vgg{item}.fwd")
            pool = i + 1 < len(VGG_TRUNK) and VGG_TRUNK[i + 1] == "M" and h.shape[1] % 2 == 0 and h.shape[2] % 2 == 0
            w4 = _vgg_wino4(w, h.shape[0], h.shape[1], h.shape[2], w.shape[0], wino4, VGG_WINO4_FWD)
            if pool and POOL_CODE and not w4 and O.conv_pool_code_supported(tuple(h.shape), w.shape[0]):
                # the full-resolution output of a pooled conv has two readers, the pool and the pool's backward: it is not written
                # at all -- the pooled tensor and a byte of (arg-max position, ReLU gate) per pooled element leave the conv instead
                pooled, code = O.conv_fwd_pool_code(h, w, V[f"{item}.bias"])
                o = None
            else:
                o = O.conv_fwd(h, w, V[f"{item}.bias"], 3, 1, 1, act=O.ACT_RELU, pool=pool, wino4=w4)
                if pool:
                    o, pooled = o
            if keep:
                steps.append(NS(kind="C", w=w, x_shape=tuple(h.shape), a=o))
        h = o
    return h, NS(steps=steps)


def vgg_backward(ctx, dfeat, nb=None, wino4=None, gated=False):
    """Input gradient only (weights are frozen, losses.py:33-34) for the first `nb` samples.
    wino4 (dgrad kernels): None = F(4x4,3x3) where it pays (TG_VGG_WINO4=0: never); True / False force.
    gated: dfeat is already the gradient in front of the trunk's last ReLU (O.l1_mean(..., relu_gate=True))."""
    da = dfeat
    steps = ctx.steps
    w4 = wino4
    for i in range(len(steps) - 1, -1, -1):
        st = steps[i]
        if st.kind == "M":
            # the pooled tensor is a ReLU output: its backward is fused into the pool backward
            if getattr(st, "code", None) is not None:
                da = O.maxpool2_bwd_code(da, st.code)        # (code of all the forward's images: the first nb are read)
            else:
                x = st.x if nb is None else st.x[:nb]
                da = O.maxpool2_bwd(da, x, relu_gate=True)
            gated = True
        else:
            a = None if st.a is None else (st.a if nb is None else st.a[:nb])      # (None: a pooled conv under the pool-code path)
            dy = da if gated else O.act_bwd(da, a, O.ACT_RELU)
            shp = st.x_shape if nb is None else (nb,) + tuple(st.x_shape[1:])
            below = steps[i - 1] if i > 0 else None
            O.tag(f"vgg{VGG_TRUNK[i]}.dgrad")
            if below is not None and below.kind == "C":      # input of this conv = ReLU output of the conv below
                ga = below.a if nb is None else below.a[:nb]
                da = O.conv_dgrad(dy, st.w, shp, 3, 1, 1, gate=ga, gate_act=O.ACT_RELU,
                                  wino4=_vgg_wino4(st.w, shp[0], shp[1], shp[2], shp[3], w4, VGG_WINO4))
                gated = True
            else:
                da = O.conv_dgrad(dy, st.w, shp, 3, 1, 1, wino4=_vgg_wino4(st.w, shp[0], shp[1], shp[2], shp[3], w4, VGG_WINO4))
                gated = False
    return da.reshape(da.shape[0], da.shape[1], da.shape[2])
