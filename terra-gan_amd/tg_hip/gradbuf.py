"""Persistent gradient storage: one flat fp32 buffer per module, every trainable parameter's gradient is a view of
it with the PARAMETER'S OWN physical layout (channels_last conv weights stay channels_last).  The backward kernels
write straight into these views, so
  * `p.grad` objects and their addresses are stable step after step (the multi-tensor Adam table is built once),
  * the data-parallel all-reduce runs on slices of the flat buffer with no packing copies,
  * nothing is allocated per step for gradients.
Parameters are laid out in REVERSE registration order (the order backward produces them), so that all-reduce buckets
cut from the front of the buffer complete first."""
import torch


ALIGN = 64          # floats


def _dense(p):
    return p.is_contiguous() or (p.dim() == 4 and p.permute(0, 2, 3, 1).is_contiguous())


class GradBuffers:
    def __init__(self, module):
        named = [(k, p) for k, p in module.named_parameters() if p.requires_grad][::-1]
        if not named:
            raise ValueError("GradBuffers: module has no trainable parameters")
        self.device = named[0][1].device
        # every view starts on a 256-byte boundary (the kernels want 16-byte aligned pointers; padding stays zero)
        total = sum((p.numel() + ALIGN - 1) // ALIGN * ALIGN for _k, p in named)
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.views, self.order, self.offsets = {}, [], {}
        off = 0
        for k, p in named:
            if not _dense(p):
                raise ValueError(f"GradBuffers: parameter {k} is not dense ({tuple(p.shape)}/{p.stride()})")
            self.views[k] = torch.as_strided(self.flat, p.shape, p.stride(), off)
            self.offsets[k] = off
            self.order.append((k, p))
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.key = tuple(p.data_ptr() for _k, p in named)

    def valid_for(self, module):
        return self.key == tuple(p.data_ptr() for _k, p in module.named_parameters() if p.requires_grad)[::-1]


def grad_buffers(module):
    """The module's GradBuffers, (re)created when its parameters moved (``.to()``, dtype change, re-registration)."""
    gb = module.__dict__.get("_tg_gradbuf")
    if gb is None or not gb.valid_for(module):
        gb = GradBuffers(module)
        module.__dict__["_tg_gradbuf"] = gb
    return gb
