"""Data-parallel gradient exchange for the train step: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" on ROCm), gloo on CPU for the logic tests.

The reference has no distributed code; the semantics implemented are those of
SURVEY §8e: every rank runs the step on its own micro-batch from identical weights, generator and
discriminator gradients are summed over ranks and divided by world_size, one Adam step follows.
BatchNorm statistics stay per-rank.

Mechanics: gradients are packed into one flat fp32 buffer per model (device-to-device copies), cut
into ~bucket_mb buckets in reverse registration order (the order backward produces them), and each
bucket is all-reduced asynchronously.  The caller then walks the buckets in order -- wait(bucket k),
run Adam on bucket k's parameters -- so the reduction of bucket k+1 overlaps the optimiser work of
bucket k.  xGMI is point-to-point (7 links/GPU); a few ~25 MB buckets keep every link busy without
serialising the optimiser behind one 103 MB collective.
"""
import torch
import torch.distributed as dist

from .gradbuf import grad_buffers


class Bucket:
    __slots__ = ("params", "flat", "work", "owner")

    def __init__(self, params, flat, owner=None):
        self.params, self.flat, self.work, self.owner = params, flat, None, owner

    def wait(self):
        if self.work is not None:
            self.work.wait()          # NCCL: makes the current stream wait; gloo: blocks the host
            self.work = None
            if self.owner is not None:
                self.owner._collective_done()


class GradSync:
    def __init__(self, world_size=None, bucket_mb=25.0, group=None, cu_reserve=None):
        """cu_reserve: CUs the one-workgroup-per-CU Winograd launches leave free so that RCCL's kernels can run underneath
        them (tg_set_cu_reserve).  Default: $TG_CU_RESERVE, else 8 (one per XCD) when world_size > 1.  The reserve is in
        force only WHILE a collective launched here is in flight (from its launch to the wait that orders the compute
        stream behind it): kernels enqueued outside that window use all CUs -- with 248 workgroups a layer of exactly 256
        work items would take two rounds.  UNMEASURED on a multi-GPU node (the build box has one GPU): the value is a
        tunable, not a tuned constant."""
        self.group = group
        self.world_size = world_size if world_size is not None else dist.get_world_size(group)
        import os
        if cu_reserve is None:
            cu_reserve = int(os.environ.get("TG_CU_RESERVE", "8" if self.world_size > 1 else "0"))
        self.cu_reserve = cu_reserve
        self._inflight = 0
        self._set_reserve(0)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self._plans = {}

    def _set_reserve(self, r):
        if torch.cuda.is_available():
            from . import lib as L
            L.check(L.load().tg_set_cu_reserve(int(r)), "tg_set_cu_reserve")

    def _collective_done(self):
        self._inflight -= 1
        if self._inflight == 0:
            self._set_reserve(0)

    @property
    def grad_scale(self):
        return 1.0 / self.world_size

    def _all_reduce(self, flat):
        """SUM `flat` over the ranks in place, asynchronously; returns a handle with .wait().  The one place that
        touches the transport (RCCL / gloo) -- tests substitute an in-process exchange here."""
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _plan(self, module, tag):
        gb = grad_buffers(module)                 # shared with the backward kernels: gradients are born in `flat`
        plan = self._plans.get(tag)
        if plan is None or plan[0] is not gb.flat:
            buckets, cur, start, end = [], [], 0, 0
            for k, p in gb.order:
                off = gb.offsets[k]
                cur.append((p, off))
                end = off + p.numel()
                if end - start >= self.bucket_elems:
                    buckets.append((cur, start, end))
                    cur, start = [], (end + 63) // 64 * 64
            if cur:
                buckets.append((cur, start, end))
            plan = (gb.flat, buckets)
            self._plans[tag] = plan
        return plan

    def __call__(self, module, tag):
        """Pack module.<param>.grad into the flat buffer, launch the bucket all-reduces, re-point every
        .grad at its (summed) slice.  Returns the buckets in launch order."""
        flat, buckets = self._plan(module, tag)
        out = []
        for plist, start, end in buckets:
            for p, off in plist:
                if p.grad is None:
                    raise RuntimeError("GradSync: a trainable parameter has no gradient")
                # slice with the parameter's own physical layout (channels_last weights stay channels_last)
                view = torch.as_strided(flat, p.shape, p.stride(), off) if _dense(p) else flat[off:off + p.numel()].view(p.shape)
                if p.grad.data_ptr() != view.data_ptr():      # gradient produced elsewhere (e.g. through autograd): pack it
                    view.copy_(p.grad)
                    p.grad = view
            b = Bucket([p for p, _ in plist], flat[start:end], self)
            if self.world_size > 1:
                if self._inflight == 0:
                    self._set_reserve(self.cu_reserve)
                self._inflight += 1
                b.work = self._all_reduce(b.flat)
            out.append(b)
        return out


def _dense(p):
    return p.is_contiguous() or (p.dim() == 4 and p.permute(0, 2, 3, 1).is_contiguous())


# --------------------------------------------------------------------------------------------------
# what a data-parallel train() needs besides the gradient exchange
# --------------------------------------------------------------------------------------------------
def rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def broadcast_state(modules=(), optimizers=(), src=0, group=None):
    """Make every rank start from rank `src`'s parameters, buffers and optimiser state ("identical weights").
    Without this, replicas built from different seeds (or resumed from different files) would silently average gradients
    taken at different weights.  Tensors are broadcast in place, so parameter storage -- and with it the gradient
    buffers and Adam tables -- stays where it is."""
    rank, world = rank_world(group)
    if world == 1:
        return
    from . import ops as O
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=group)
        O.weights_updated(list(m.parameters()))       # written behind torch's version counter: prepared conv weights are stale
    for opt in optimizers:
        # state may be empty on some ranks only if it is empty on all (a fresh optimiser): exchange the structure first
        meta = [opt.state_dict()] if rank == src else [None]
        dist.broadcast_object_list(meta, src=src, group=group)
        if rank != src and meta[0]["state"]:
            opt.load_state_dict(meta[0])       # Optimizer.load_state_dict moves the state to each parameter's device


class ShardSampler(torch.utils.data.Sampler):
    """Per-rank index split of a dataset (DistributedSampler semantics): one shared permutation per epoch (seed + epoch),
    padded by wrap-around to a multiple of world so every rank draws the same number of batches -- the gradient all-reduces
    of the ranks pair up one to one."""

    def __init__(self, n, rank, world, shuffle=True, seed=0):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0
        self.per_rank = (n + world - 1) // world

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.per_rank

    def __iter__(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        total = self.per_rank * self.world
        while len(idx) < total:
            idx += idx[:total - len(idx)]
        return iter(idx[self.rank:total:self.world])
