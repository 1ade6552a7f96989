"""Shared helpers of the model mirrors: tensor dictionaries for the HIP engines."""
import torch

from tg_hip import lib as _lib


def require_hip(t, who):
    if not t.is_cuda:
        raise _lib.TgError(f"{who}: input is on {t.device}; the MI355X build has no CPU path "
                           "(move the module and its inputs to cuda)")
    _lib.load()


class TensorDictMixin:
    """Caches {state-dict key: tensor} for parameters and buffers; dropped whenever the module is
    moved / cast / re-loaded so the engines always see the live storage."""

    def _tensors(self):
        cache = self.__dict__.get("_tg_cache")
        if cache is None:
            cache = dict(self.named_parameters())
            cache.update(dict(self.named_buffers()))
            self.__dict__["_tg_cache"] = cache
        return cache

    def _apply(self, fn, *a, **kw):
        self.__dict__.pop("_tg_cache", None)
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self.__dict__.pop("_tg_cache", None)
        return super().load_state_dict(*a, **kw)

    def _trainable(self):
        return [(k, p) for k, p in self.named_parameters() if p.requires_grad]


def to_channels_last_(conv):
    """Store an nn.Conv2d weight as [Cout][kh][kw][Cin] (logical shape and values unchanged)."""
    w = conv.weight
    w.data = w.data.contiguous(memory_format=torch.channels_last)


def as_bhw(x, who):
    """[B,1,H,W] -> contiguous [B,H,W] view."""
    if x.dim() != 4 or x.shape[1] != 1:
        raise ValueError(f"{who}: expected a [B,1,H,W] tensor, got {tuple(x.shape)}")
    x = x.detach()
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous().reshape(x.shape[0], x.shape[2], x.shape[3])
