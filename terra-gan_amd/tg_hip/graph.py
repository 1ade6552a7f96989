"""A whole GAN train step captured once in a hipGraph and replayed.

Why: at small batches the step is bound by its ~340 kernel launches, not by the kernels (256x256 / B=1: 4.6 ms per
step for ~1.6 ms of kernel time; the reference ships batch 2, train.py:77).  A captured step replays all launches from
one hipGraphLaunch: no per-launch host work, back-to-back dispatch on the device.

What makes the step capturable: nothing in `train_step` synchronises with the host, every buffer the kernels touch is
either persistent (parameters, gradient buffers, Adam state, prepared weights, workspaces) or comes from torch's
graph-private memory pool, and the only kernel arguments that change from step to step -- Adam's two bias-correction
scalars -- are read from device memory (tg_adam_multi_s) and refreshed before each replay (ops.AdamScalarArena).
Results are bit-identical to the eager step (tests/test_hip_graph.py).

Single-GPU only: the data-parallel path keeps the eager schedule (its collectives are launched by torch.distributed).
"""
import torch

from . import ops as O


class GraphedTrainStep:
    """step = GraphedTrainStep(G, D, criterion, optG, optD); out = step(real, mask)

    The first `warmup` calls run eagerly (they create the optimiser state, gradient buffers, workspaces, prepared-weight
    tables and per-kernel LDS opt-ins -- none of which may happen inside a capture); the next call captures; every call
    from then on copies the batch into the static input buffers and replays.  The returned dict holds STATIC tensors that
    the next call overwrites.  `flush()` brings the torch optimisers' host-side step counters up to date (state_dict(),
    checkpoints); it is cheap and idempotent."""

    def __init__(self, generator, discriminator, criterion, optimizer_G, optimizer_D, warmup=2, reuse_fake_forward=True):
        self.G, self.D, self.crit, self.oG, self.oD = generator, discriminator, criterion, optimizer_G, optimizer_D
        self.warmup, self.reuse = max(int(warmup), 1), reuse_fake_forward
        self.calls, self.graph, self.out, self.shape = 0, None, None, None
        self.replays, self.flushed = 0, 0
        self.arena = None

    def _eager(self, real, mask):
        from mvp_gan.src.train import train_step
        return train_step(self.G, self.D, self.crit, self.oG, self.oD, real, mask, reuse_fake_forward=self.reuse)

    def _capture(self, real, mask):
        dev = real.device
        self.real_s, self.mask_s = torch.empty_like(real), torch.empty_like(mask)
        self.shape = (tuple(real.shape), tuple(mask.shape))
        self.arena = O.AdamScalarArena(dev)
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        O.adam_scalar_arena = self.arena
        self.arena.active = True
        O.wprep_capture_log = log = []
        try:
            with torch.cuda.graph(self.graph):
                self.out = self._eager(self.real_s, self.mask_s)
        finally:
            self.arena.active = False
            O.adam_scalar_arena = None
            O.wprep_capture_log = None
        # what a replay changes behind the host-side prepared-weight cache (ops.graph_replayed)
        self.wprep_keys = list(dict.fromkeys(log))
        self.wptrs = sorted({p.data_ptr() for m in (self.G, self.D) for p in m.parameters() if p.dim() == 4 and p.requires_grad})
        # the capture executed hip_adam_step's HOST side (state["step"] += 1) without running a step on the device
        for opt in (self.oG, self.oD):
            for st in opt.state.values():
                if "step" in st:
                    st["step"] -= 1

    def __call__(self, real, mask):
        self.calls += 1
        if self.graph is None:
            if self.calls <= self.warmup:
                return self._eager(real, mask)
            self._capture(real, mask)
        if (tuple(real.shape), tuple(mask.shape)) != self.shape:
            raise ValueError(f"GraphedTrainStep: captured for batch shape {self.shape[0]}, got {tuple(real.shape)}")
        self.real_s.copy_(real, non_blocking=True)
        self.mask_s.copy_(mask, non_blocking=True)
        self.arena.refresh(self.replays)
        self.graph.replay()
        self.replays += 1
        O.graph_replayed(self.wptrs, self.wprep_keys)
        return self.out

    def flush(self):
        n = self.replays - self.flushed
        if n:
            for opt in (self.oG, self.oD):
                for st in opt.state.values():
                    if "step" in st:
                        st["step"] += n
            self.flushed = self.replays
