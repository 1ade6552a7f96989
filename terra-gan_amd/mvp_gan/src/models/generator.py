"""PConvUNet -- the 7-down / 7-up partial-convolution U-Net generator on MI355X.

Constructor (no arguments), `forward(x, mask) -> output`, children names and state-dict keys match
/root/reference/mvp_gan/src/models/generator.py:8-84; layers are created in the reference's order
(enc1..enc7, dec7..dec1, final) so seeded construction gives identical parameters.  The whole
forward (mask pyramid, 14 partial convs, bilinear-up/concat, final conv, sigmoid composite) and its
hand-scheduled backward run in tg_hip.engine on HIP kernels; autograd sees one node.
"""
import torch
import torch.nn as nn

from tg_hip import engine as E

from ._common import TensorDictMixin, as_bhw, require_hip, to_channels_last_
from .pconv import PConv2d


class _GeneratorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, mod, *params):
        xb, mb = as_bhw(x, "PConvUNet"), as_bhw(mask, "PConvUNet")
        out, c = E.generator_forward(mod._tensors(), xb, mb, mod.training)
        if any(ctx.needs_input_grad):
            ctx.c, ctx.mod = c, mod
        return out.reshape(x.shape)

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        want_dx = ctx.needs_input_grad[0]
        grads, dx = E.generator_backward(mod._tensors(), ctx.c, as_bhw(dout, "PConvUNet.backward").clone(), want_dx)
        ctx.c = None
        out = [dx.reshape(dout.shape) if dx is not None else None, None, None]
        out += [grads[k] for k, _p in mod._trainable()]
        return tuple(out)


class PConvUNet(TensorDictMixin, nn.Module):
    def __init__(self):
        super().__init__()
        for name, cin, cout, k, s, p in E.G_ENC + E.G_DEC:
            setattr(self, name, PConv2d(cin, cout, kernel_size=k, stride=s, padding=p))
        self.final = nn.Conv2d(64, 1, kernel_size=3, padding=1)
        to_channels_last_(self.final)
        # train_step(): drop post-activation / concat tensors after forward and recompute them in backward
        # (BASELINE config 5).  Off by default: 288 GB of HBM hold 1024x1024 tiles at batch 4 without it.
        self.activation_checkpointing = False

    def forward(self, x, mask):
        require_hip(x, "PConvUNet")
        return _GeneratorFn.apply(x, mask, self, *[p for _k, p in self._trainable()])
