"""Discriminator -- 5-conv PatchGAN on MI355X.

`Discriminator(input_channels=1)`, `forward(img)`, the `model` Sequential and its state-dict keys
(model.{0,2,5,8,11}.*, model.{3,6,9}.*) match /root/reference/mvp_gan/src/models/discriminator.py:6-26.
The Sequential only holds parameters; forward/backward run in tg_hip.engine.
"""
import torch
import torch.nn as nn

from tg_hip import engine as E
from tg_hip import ops as O

from ._common import TensorDictMixin, require_hip, to_channels_last_


class _DiscriminatorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, mod, *params):
        x = O.nchw_to_nhwc(img.detach().float())
        logits, c = E.discriminator_forward(mod._tensors(), x, mod.training)
        if any(ctx.needs_input_grad):
            ctx.c, ctx.mod = c, mod
        return O.nhwc_to_nchw(logits)

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        want_w = any(ctx.needs_input_grad[2:])
        dl = O.nchw_to_nhwc(dout.float())
        if dl.data_ptr() == dout.data_ptr():
            dl = dl.clone()
        grads, dimg = E.discriminator_backward(mod._tensors(), ctx.c, dl, want_wgrad=want_w,
                                               want_dimg=ctx.needs_input_grad[0])
        ctx.c = None
        out = [O.nhwc_to_nchw(dimg) if dimg is not None else None, None]
        out += [grads.get(k) for k, _p in mod._trainable()]
        return tuple(out)


class Discriminator(TensorDictMixin, nn.Module):
    def __init__(self, input_channels=1):
        super().__init__()
        layers, cin = [], input_channels
        for i, cout in enumerate((64, 128, 256, 512)):
            layers.append(nn.Conv2d(cin, cout, kernel_size=4, stride=2, padding=1))
            if i > 0:
                layers.append(nn.BatchNorm2d(cout))
            layers.append(nn.LeakyReLU(0.2, inplace=True))
            cin = cout
        layers.append(nn.Conv2d(512, 1, kernel_size=4, padding=1))
        self.model = nn.Sequential(*layers)
        for m in self.model:
            if isinstance(m, nn.Conv2d):
                to_channels_last_(m)

    def forward(self, img):
        require_hip(img, "Discriminator")
        return _DiscriminatorFn.apply(img, self, *[p for _k, p in self._trainable()])
