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
    __slots__ = ("params", "flat", "work", "owner", "keys")

    def __init__(self, params, flat, owner=None, keys=()):
        self.params, self.flat, self.work, self.owner, self.keys = params, flat, None, owner, keys

    def wait(self):
        if self.work is not None:
            self.work.wait()          # NCCL: makes the current stream wait; gloo: blocks the host
            self.work = None
            if self.owner is not None:
                self.owner._collective_done()


class BucketLauncher:
    """Launches each bucket's all-reduce as soon as the LAST gradient of the bucket has been enqueued:
    the backward schedule calls ready(key) after the kernel writing that parameter's gradient was launched; NCCL/RCCL
    orders the collective behind everything enqueued on the compute stream at that moment, so the reduction of the
    early buckets (final, dec1, dec2 ...) runs underneath the rest of the backward.  finish() launches whatever is
    left (parameters whose gradient never became ready individually) and returns the buckets in launch order."""

    def __init__(self, sync, module, tag):
        self.sync, self.module = sync, module
        self.flat, self.plan = sync._plan(module, tag)
        self.pending = [set(k for _p, _off, k in plist) for plist, _s, _e in self.plan]
        self.where = {k: i for i, (plist, _s, _e) in enumerate(self.plan) for _p, _off, k in plist}
        self.param = {k: p_ for plist, _s, _e in self.plan for p_, _off, k in plist}
        self.buckets = [None] * len(self.plan)
        self.launched = []

    def ready(self, key, grad=None):
        i = self.where.get(key)
        if i is None or self.buckets[i] is not None:
            return
        if grad is not None:
            self.param[key].grad = grad
        self.pending[i].discard(key)
        if not self.pending[i]:
            self._launch(i)

    def _launch(self, i):
        plist, start, end = self.plan[i]
        self.sync._pack(self.flat, plist)
        b = Bucket([p for p, _off, _k in plist], self.flat[start:end], self.sync, tuple(k for _p, _off, k in plist))
        self.sync._start(b)
        self.buckets[i] = b
        self.launched.append(b)

    def finish(self):
        for i in range(len(self.plan)):
            if self.buckets[i] is None:
                self._launch(i)
        # A bucket that went out early was packed at ITS launch; a caller that pointed p.grad somewhere else afterwards
        # (e.g. re-assigned the unreduced tensors an engine returned without persistent gradient buffers) would make the
        # optimiser read unreduced gradients scaled by 1/world.  Refuse that pairing instead of training on it.
        for plist, _s, _e in self.plan:
            for p, off, k in plist:
                if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                    raise RuntimeError(f"GradSync: the gradient of '{k}' no longer aliases the reduced bucket -- a backward that "
                                       "reports gradients with ready(key, grad) must write them into tg_hip.gradbuf views "
                                       "(or must not re-assign .grad after reporting them)")
        return self.launched


class GradSync:
    def __init__(self, world_size=None, bucket_mb=25.0, group=None, cu_reserve=None):
        """cu_reserve: CUs the one-workgroup-per-CU Winograd launches leave free so that RCCL's kernels can run underneath
        them (tg_set_cu_reserve).  Default: $TG_CU_RESERVE, else 0 -- OPT-IN: no multi-GPU run has shown yet that a reserve
        helps (the build box has one GPU).  When set, the reserve is in force only WHILE a collective launched here is in
        flight (from its launch to the wait that orders the compute stream behind it), and it changes the GRID of the
        persistent kernels only -- split-K plans (summation order) are always made for 256 CUs, so the numbers are those
        of the single-GPU path."""
        self.group = group
        self.world_size = world_size if world_size is not None else dist.get_world_size(group)
        import os
        if cu_reserve is None:
            cu_reserve = int(os.environ.get("TG_CU_RESERVE", "0"))
        self.cu_reserve = cu_reserve
        self._inflight = 0
        self._set_reserve(0)
        if self.world_size > 1 and torch.cuda.is_available():
            # the one-workgroup-per-CU Winograd launches pull their work items from queues (tg_set_work_stealing): a workgroup
            # whose CU is held by one of RCCL's long-running kernels then delays nothing but itself.  Same bits as the static walk.
            from . import lib as L
            L.check(L.load().tg_set_work_stealing(1), "tg_set_work_stealing")
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self._plans = {}

    def _set_reserve(self, r):
        if self.cu_reserve and torch.cuda.is_available():
            from . import lib as L
            L.check(L.load().tg_set_cu_reserve(int(r)), "tg_set_cu_reserve")

    def _collective_done(self):
        self._inflight = max(0, self._inflight - 1)
        if self._inflight == 0:
            self._set_reserve(0)

    def reset(self):
        """Forget collectives in flight (a step raised between launch and wait): the reserve goes back to 0."""
        self._inflight = 0
        self._set_reserve(0)

    @property
    def grad_scale(self):
        return 1.0 / self.world_size

    def _all_reduce(self, flat):
        """SUM `flat` over the ranks in place, asynchronously; returns a handle with .wait() -- or None when the exchange
        completed synchronously.  The one place that touches the transport (RCCL / gloo) -- tests substitute an
        in-process exchange here."""
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _plan(self, module, tag):
        gb = grad_buffers(module)                 # shared with the backward kernels: gradients are born in `flat`
        plan = self._plans.get(tag)
        if plan is None or plan[0] is not gb.flat:
            buckets, cur, start, end = [], [], 0, 0
            for k, p in gb.order:
                off = gb.offsets[k]
                cur.append((p, off, k))
                end = off + p.numel()
                if end - start >= self.bucket_elems:
                    buckets.append((cur, start, end))
                    cur, start = [], (end + 63) // 64 * 64
            if cur:
                buckets.append((cur, start, end))
            plan = (gb.flat, buckets)
            self._plans[tag] = plan
        return plan

    @staticmethod
    def _pack(flat, plist):
        for p, off, _k in plist:
            if p.grad is None:
                raise RuntimeError("GradSync: a trainable parameter has no gradient")
            # slice with the parameter's own physical layout (channels_last weights stay channels_last)
            view = torch.as_strided(flat, p.shape, p.stride(), off) if _dense(p) else flat[off:off + p.numel()].view(p.shape)
            if p.grad.data_ptr() != view.data_ptr():      # gradient produced elsewhere (e.g. through autograd): pack it
                view.copy_(p.grad)
                p.grad = view

    def _start(self, b):
        if self.world_size > 1:
            if self._inflight == 0:
                self._set_reserve(self.cu_reserve)
            self._inflight += 1
            try:
                b.work = self._all_reduce(b.flat)
            except BaseException:
                self.reset()
                raise
            if b.work is None:                     # synchronous transport: nothing is in flight
                self._collective_done()

    def begin(self, module, tag):
        """Eager mode: returns a BucketLauncher whose ready(key) the backward schedule calls per finished gradient."""
        return BucketLauncher(self, module, tag)

    def __call__(self, module, tag):
        """Pack module.<param>.grad into the flat buffer, launch the bucket all-reduces, re-point every
        .grad at its (summed) slice.  Returns the buckets in launch order."""
        return BucketLauncher(self, module, tag).finish()


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
        # Only the STRUCTURE travels as an object (param_groups, step counts, which parameters have state); the moments are
        # created on each rank with the parameter's own layout and broadcast in place -- pickling rank 0's CUDA tensors would
        # materialise every replica's copy on rank 0's device first (7 extra contexts on an 8-GPU resume).
        params = [p for g in opt.param_groups for p in g["params"]]
        if rank == src:
            sd = opt.state_dict()
            # which entries are tensors is decided by KEY / rank, never by size: the moments of a 1-element parameter
            # (G final.bias, D model.11.bias) are tensors like any other; only `step` (a 0-dim tensor) travels as a number
            meta = [{"param_groups": sd["param_groups"],
                     "state": {i: {k: _state_meta(k, v) for k, v in st.items()} for i, st in sd["state"].items()}}]
        else:
            meta = [None]
        dist.broadcast_object_list(meta, src=src, group=group)
        meta = meta[0]
        if not meta["state"]:
            continue
        if rank != src:
            for g, mg in zip(opt.param_groups, meta["param_groups"]):
                g.update({k: v for k, v in mg.items() if k != "params"})
        for i, p in enumerate(params):
            ms = meta["state"].get(i)
            if ms is None:
                continue
            st = opt.state[p]
            for k, v in ms.items():
                if v is None:                                  # a tensor-valued entry (exp_avg, exp_avg_sq ...)
                    if rank != src or k not in st:
                        if k not in st or st[k].shape != p.shape:
                            st[k] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    t = st[k]
                    if not ((t.is_contiguous() and p.is_contiguous()) or t.stride() == p.stride()):
                        st[k] = t = torch.empty_like(p, memory_format=torch.preserve_format).copy_(t)
                    dist.broadcast(t, src=src, group=group)
                elif k == "step":
                    st[k] = torch.tensor(float(v), dtype=torch.float32)
                else:
                    st[k] = v


def _state_meta(key, v):
    """What broadcast_state ships as an object for one optimiser-state entry: None = "a tensor, broadcast in place"."""
    if not torch.is_tensor(v):
        return v
    if key == "step" or v.dim() == 0:
        return float(v)
    return None


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
