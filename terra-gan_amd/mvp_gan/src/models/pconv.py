"""PConv2d -- partial convolution + mask update + renormalisation + BatchNorm + ReLU on MI355X.

Same constructor, forward signature, children and state-dict keys as the reference layer
(/root/reference/mvp_gan/src/models/pconv.py:6-50); the arithmetic runs as HIP kernels
(tg_mask_update, tg_conv_fwd/dgrad/wgrad, tg_bn_*).  `input_conv`, `mask_conv` and `bn` are kept
as ordinary nn children purely as parameter holders, built in the reference's order so that
`torch.manual_seed(s); PConv2d(...)` yields bit-identical parameters (pconv.py:9-21).
"""
import torch
import torch.nn as nn

from tg_hip import engine as E
from tg_hip import ops as O

from ._common import TensorDictMixin, require_hip, to_channels_last_


class _PConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, mod, *params):
        P = mod._tensors()
        xh = O.nchw_to_nhwc(x.detach().float())
        m = mask.detach().float().contiguous().reshape(mask.shape[0], mask.shape[2], mask.shape[3])
        k, s, p = mod.kernel_size, mod.stride, mod.padding
        mo, ratio = O.mask_update(m, k, s, p)                                     # pconv.py:33-40
        if mod.batch_norm:
            a, c = E._pconv_fwd(_Prefixed(P), "L", k, s, p, xh, m, ratio, mod.training)
        else:                                                                      # pconv.py:46-48 without bn
            a = O.conv_fwd(xh, P["input_conv.weight"], P["input_conv.bias"].detach(), k, s, p, in_mask=m, ratio=ratio,
                           act=O.ACT_RELU)
            c = E.NS(name="L", k=k, s=s, p=p, x=xh, in_mask=m, ratio=ratio, a=a)
        ctx.c, ctx.mod = c, mod
        ctx.mark_non_differentiable(mo)
        mo4 = mo.reshape(mo.shape[0], 1, mo.shape[1], mo.shape[2])
        ctx.mark_non_differentiable(mo4)
        return O.nhwc_to_nchw(a), mo4

    @staticmethod
    def backward(ctx, dout, _dmask):
        mod, c = ctx.mod, ctx.c
        P = mod._tensors()
        da = O.nchw_to_nhwc(dout.float())
        if da.data_ptr() == dout.data_ptr():
            da = da.clone()
        grads = {}
        want_dx = ctx.needs_input_grad[0]
        if mod.batch_norm:
            dx = E._pconv_bwd(_Prefixed(P), c, da, grads, want_dx=want_dx)
        else:
            w = P["input_conv.weight"]
            dyr = O.act_bwd(da, c.a, O.ACT_RELU, ratio=c.ratio)
            grads["L.input_conv.weight"], grads["L.input_conv.bias"] = O.conv_wgrad(c.x, dyr, w, c.k, c.s, c.p, in_mask=c.in_mask)
            dx = O.conv_dgrad(dyr, w, tuple(c.x.shape), c.k, c.s, c.p, in_mask=c.in_mask) if want_dx else None
        out = [O.nhwc_to_nchw(dx) if dx is not None else None, None, None]
        out += [grads.get("L." + k_) for k_, _p in mod._trainable()]
        return tuple(out)


class _Prefixed(dict):
    """View of a layer-local tensor dict under the name 'L.' expected by the engine helpers."""

    def __init__(self, base):
        super().__init__({"L." + k: v for k, v in base.items()})


class PConv2d(TensorDictMixin, nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, batch_norm=True):
        super().__init__()
        self.input_conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True)
        self.slide_winsize = kernel_size * kernel_size          # 1-channel mask window (pconv.py:10)
        self.mask_conv = nn.Conv2d(1, 1, kernel_size, stride, padding, bias=False)
        nn.init.constant_(self.mask_conv.weight, 1.0)
        self.mask_conv.weight.requires_grad = False
        self.batch_norm = batch_norm
        if batch_norm:
            self.bn = nn.BatchNorm2d(out_channels)
        self.activation = nn.ReLU()
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        to_channels_last_(self.input_conv)

    def forward(self, input, mask):
        require_hip(input, "PConv2d")
        return _PConvFn.apply(input, mask, self, *[p for _k, p in self._trainable()])
