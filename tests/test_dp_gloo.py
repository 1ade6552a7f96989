"""Data-parallel logic on CPU: 2 gloo ranks exercise tg_hip.dist.GradSync (flat packing, bucketing,
async all-reduce, per-bucket wait order) on CPU tensors; the arithmetic (sum over ranks, 1/world scale,
channels_last slices) is checked against a single-process reference.  The HIP kernels are not involved."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    m = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 4, 1), nn.Linear(4, 2))
    m[0].weight.data = m[0].weight.data.contiguous(memory_format=torch.channels_last)
    return m


def _grads(rank, m):
    g = torch.Generator().manual_seed(100 + rank)
    return [torch.randn(p.shape, generator=g) for p in m.parameters()]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "terra-gan_amd"))
    from tg_hip.dist import GradSync
    m = _model()
    for p, g in zip(m.parameters(), _grads(rank, m)):
        p.grad = torch.empty_like(p).copy_(g)
    sync = GradSync(world, bucket_mb=0.00005)            # tiny buckets -> several of them
    buckets = sync(m, "M")
    assert len(buckets) >= 3
    seen = []
    for b in buckets:
        b.wait()
        seen += b.params
    assert len(seen) == len(list(m.parameters())) and seen[0] is list(m.parameters())[-1]     # reverse order
    out = [p.grad.clone() * sync.grad_scale for p in m.parameters()]
    assert m[0].weight.grad.permute(0, 2, 3, 1).is_contiguous()                              # layout kept
    # second call re-uses the flat buffer and must not accumulate stale values
    for p, g in zip(m.parameters(), _grads(rank, m)):
        p.grad = torch.empty_like(p).copy_(g)
    for b in sync(m, "M"):
        b.wait()
    out2 = [p.grad.clone() * sync.grad_scale for p in m.parameters()]
    # eager mode: a bucket's all-reduce is launched when its LAST gradient is reported ready, in whatever order
    # the gradients arrive; finish() launches the rest.  Gradients are handed over with ready(key, grad).
    from tg_hip.gradbuf import grad_buffers
    grads = dict(zip([k for k, _ in m.named_parameters()], _grads(rank, m)))
    for p in m.parameters():
        p.grad = None
    la = sync.begin(m, "M")
    order = [k for k, _p in grad_buffers(m).order]               # reverse registration order = backward order
    n_launched = []
    for k in order[:-1]:
        la.ready(k, torch.empty_like(dict(m.named_parameters())[k]).copy_(grads[k]))
        n_launched.append(len(la.launched))
    assert n_launched[-1] >= 2 and n_launched == sorted(n_launched) and n_launched[0] <= 1, n_launched
    assert sync._inflight == len(la.launched) > 0
    last = dict(m.named_parameters())[order[-1]]
    last.grad = torch.empty_like(last).copy_(grads[order[-1]])    # never reported: finish() must pick it up
    bs = la.finish()
    assert len(bs) == len(buckets) and [b.keys for b in bs] == [b.keys for b in buckets]
    for b in bs:
        b.wait()
    assert sync._inflight == 0
    out3 = [p.grad.clone() * sync.grad_scale for p in m.parameters()]
    # a gradient re-pointed AFTER its bucket went out (e.g. a caller re-assigning the unreduced tensors an engine returned) would
    # make the optimiser read unreduced values scaled by 1/world: finish() refuses that pairing
    for p in m.parameters():
        p.grad = None
    lb = sync.begin(m, "M")
    for k in order:
        lb.ready(k, torch.empty_like(dict(m.named_parameters())[k]).copy_(grads[k]))
    victim = dict(m.named_parameters())[order[0]]
    victim.grad = victim.grad.clone()
    try:
        lb.finish()
        raise AssertionError("expected finish() to refuse a re-pointed gradient")
    except RuntimeError as e:
        assert "no longer aliases" in str(e)
    for b in lb.launched:
        b.wait()
    assert sync._inflight == 0
    # a transport that completes synchronously (returns None) leaves nothing in flight; a raising one resets the count
    class SyncT(GradSync):
        def _all_reduce(self, flat):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            return None
    s2 = SyncT(world, bucket_mb=0.00005)
    for p, g in zip(m.parameters(), _grads(rank, m)):
        p.grad = torch.empty_like(p).copy_(g)
    for b in s2(m, "M"):
        assert s2._inflight == 0
        b.wait()
    out4 = [p.grad.clone() * s2.grad_scale for p in m.parameters()]
    class BadT(GradSync):
        def _all_reduce(self, flat):
            raise RuntimeError("transport down")
    s3 = BadT(world, bucket_mb=0.00005)
    try:
        s3(m, "M")
        raise AssertionError("expected the transport error")
    except RuntimeError:
        assert s3._inflight == 0
    q.put((rank, [t.numpy().copy() for t in out], [t.numpy().copy() for t in out2], [t.numpy().copy() for t in out3],
           [t.numpy().copy() for t in out4]))   # plain arrays: no shm handles
    dist.destroy_process_group()


def test_gradsync_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = _model()
    want = [sum(gs) / world for gs in zip(*[_grads(r, m) for r in range(world)])]
    for _rank, out, out2, out3, out4 in res:
        for a, b, c, d, w in zip(out, out2, out3, out4, want):
            a, b, c, d = torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(c), torch.from_numpy(d)
            assert torch.allclose(a, w, atol=1e-6) and torch.allclose(b, w, atol=1e-6)
            assert torch.equal(c, a) and torch.equal(d, a)        # eager / synchronous transports: the same bits


def test_shard_sampler_partitions_every_epoch():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "terra-gan_amd"))
    from tg_hip.dist import ShardSampler
    for n, world in [(10, 2), (7, 4), (16, 8), (3, 8)]:
        for epoch in range(3):
            parts = []
            for r in range(world):
                sm = ShardSampler(n, r, world, shuffle=True, seed=5)
                sm.set_epoch(epoch)
                parts.append(list(sm))
            assert len({len(p) for p in parts}) == 1 and len(parts[0]) == (n + world - 1) // world   # equal batch counts
            flat = [i for p in parts for i in p]
            assert set(flat) == set(range(n)) or n < world and set(flat) <= set(range(n))     # every sample is drawn
            if n >= world:
                assert len(flat) - len(set(flat)) == len(parts[0]) * world - n              # only the wrap-around padding repeats
        a, b = ShardSampler(n, 0, world, seed=5), ShardSampler(n, 0, world, seed=5)
        b.set_epoch(1)
        if n >= 2 * world:
            assert list(a) != list(b)                                                        # reshuffled per epoch


def _bcast_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "terra-gan_amd"))
    from tg_hip.dist import broadcast_state
    torch.manual_seed(rank)                           # different weights per rank
    # the last conv has a ONE-element bias, like G's final.bias (Conv2d(64,1)) and D's model.11.bias (Conv2d(512,1)): its Adam
    # moments are 1-element tensors and must stay tensors (hip_adam_step takes their data_ptr)
    m = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8), nn.Conv2d(8, 1, 3))
    m[0].weight.data = m[0].weight.data.contiguous(memory_format=torch.channels_last)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    if rank == 0:                                     # only rank 0 has optimiser state (e.g. resumed from a checkpoint)
        for p in m.parameters():
            p.grad = torch.ones_like(p)
        opt.step()
        m[1].running_mean.fill_(0.25)
    ptr = m[0].weight.data_ptr()
    broadcast_state([m], [opt])
    assert m[0].weight.data_ptr() == ptr              # in place: gradient buffers / Adam tables stay valid
    st = opt.state[m[0].weight]
    for p in m.parameters():
        sp = opt.state[p]
        assert torch.is_tensor(sp["step"]) and sp["step"].dim() == 0
        for mk in ("exp_avg", "exp_avg_sq"):
            assert torch.is_tensor(sp[mk]) and sp[mk].shape == p.shape, (mk, type(sp[mk]), tuple(p.shape))
    for p in m.parameters():                          # and the state is usable: one more optimiser step on every rank
        p.grad = torch.full_like(p, 0.5)
    opt.step()
    sb = opt.state[m[2].bias]
    q.put((rank, m[0].weight.detach().numpy().copy(), m[1].running_mean.numpy().copy(), float(st["step"]),
           st["exp_avg"].numpy().copy(), sb["exp_avg"].numpy().copy(), sb["exp_avg_sq"].numpy().copy(),
           m[2].bias.detach().numpy().copy()))
    dist.destroy_process_group()


def test_broadcast_state_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(res[0], res[1]):
        assert (torch.as_tensor(a) == torch.as_tensor(b)).all()
    assert res[1][2] == 2.0 and float(res[1][1][0]) == 0.25
