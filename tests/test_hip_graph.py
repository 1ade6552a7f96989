"""hipGraph replay of the train step (tg_hip.graph.GraphedTrainStep) is bit-identical to the eager step: losses, generated
batch, every weight, BatchNorm buffer and Adam moment after several steps -- at the reference's own CPU-runnable case
(256x256, batch 1: BASELINE configs[0]) and at its shipped batch size 2 (train.py:77)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tg_hip import lib
    lib.load()
    return torch.device("cuda:0")


def _build(dev):
    from mvp_gan.src.models import Discriminator, PConvUNet
    from mvp_gan.src.utils.losses import InpaintingLoss
    torch.manual_seed(0)
    G, D = PConvUNet(), Discriminator()
    crit = InpaintingLoss(0.1, 0.1, device=torch.device("cpu"))
    G, D, crit = G.to(dev), D.to(dev), crit.to(dev)
    return G, D, crit, torch.optim.Adam(G.parameters(), lr=2e-4), torch.optim.Adam(D.parameters(), lr=2e-4)


@pytest.mark.parametrize("b,size", [(1, 256), (2, 128)])
def test_graphed_step_is_bit_identical_to_eager(dev, b, size):
    from mvp_gan.src.train import train_step
    from oracle import terragan_oracle as Orc
    from tg_hip.graph import GraphedTrainStep
    nsteps = 6
    data = [tuple(t.to(dev) for t in Orc.synth_batch(b, size, 50 + s)) for s in range(nsteps)]
    keys = ("g_total", "g_loss", "g_adv", "d_loss", "real_loss", "fake_loss")
    runs = {}
    for mode in ("eager", "graph"):
        G, D, crit, oG, oD = _build(dev)
        step = GraphedTrainStep(G, D, crit, oG, oD, warmup=2) if mode == "graph" else None
        losses, gens = [], []
        for real, mask in data:
            out = step(real, mask) if step is not None else train_step(G, D, crit, oG, oD, real, mask)
            losses.append(tuple(float(out[k]) for k in keys))          # read before the next replay overwrites the statics
            gens.append(out["gen"].detach().clone())
        if step is not None:
            assert step.replays == nsteps - 2
            step.flush()
        torch.cuda.synchronize()
        runs[mode] = (losses, gens, {k: v.detach().clone() for m in (G, D) for k, v in m.state_dict().items()},
                      [(int(st["step"]), st["exp_avg"].clone(), st["exp_avg_sq"].clone()) for o in (oG, oD) for st in o.state.values()])
    assert runs["eager"][0] == runs["graph"][0], (runs["eager"][0], runs["graph"][0])
    for a, c in zip(runs["eager"][1], runs["graph"][1]):
        assert torch.equal(a, c)
    for k, v in runs["eager"][2].items():
        assert torch.equal(v, runs["graph"][2][k]), k
    for (s1, m1, v1), (s2, m2, v2) in zip(runs["eager"][3], runs["graph"][3]):
        assert s1 == s2 == nsteps and torch.equal(m1, m2) and torch.equal(v1, v2)


def test_graphed_step_rejects_another_batch_shape(dev):
    from oracle import terragan_oracle as Orc
    from tg_hip.graph import GraphedTrainStep
    G, D, crit, oG, oD = _build(dev)
    step = GraphedTrainStep(G, D, crit, oG, oD, warmup=1)
    real, mask = (t.to(dev) for t in Orc.synth_batch(2, 128, 3))
    for _ in range(3):
        step(real, mask)
    real2, mask2 = (t.to(dev) for t in Orc.synth_batch(3, 128, 3))
    with pytest.raises(ValueError):
        step(real2, mask2)


def test_eager_ops_between_replays_see_the_replayed_weights(dev):
    """A replay rewrites G / D weights on the device; the host-side prepared-weight cache must not serve transforms of older
    weights to eager work done between replays -- the reference validates between train steps (train.py:278-301: G.eval()
    forward at the validation batch's size, D in train mode).  Compared with the same sequence run with the cache off."""
    from oracle import terragan_oracle as Orc
    from tg_hip import ops as O
    from tg_hip.graph import GraphedTrainStep
    from mvp_gan.src.train import validation_step
    data = [tuple(t.to(dev) for t in Orc.synth_batch(2, 128, 70 + s)) for s in range(5)]
    vdata = [tuple(t.to(dev) for t in Orc.synth_batch(3, 96, 90 + s)) for s in range(2)]          # another size AND batch
    runs = {}
    for cache in (True, False):
        O.WPREP_CACHE = cache
        try:
            G, D, crit, oG, oD = _build(dev)
            step = GraphedTrainStep(G, D, crit, oG, oD, warmup=1) if cache else None
            from mvp_gan.src.train import train_step
            rec = []
            for i, (real, mask) in enumerate(data):
                out = step(real, mask) if step is not None else train_step(G, D, crit, oG, oD, real, mask)
                rec.append(float(out["g_total"]))
                if i >= 1:                                   # from the first replay on: eager validation in between
                    G.eval()
                    for vr, vm in vdata:
                        gt, dl = validation_step(G, D, crit, vr, vm)
                        rec += [float(gt), float(dl)]
                    with torch.no_grad():
                        rec.append(float(D(data[0][0]).sum()))         # D forward at the TRAIN size: the non-batchable entries
                    G.train()
            torch.cuda.synchronize()
            if step is not None:
                assert step.replays == len(data) - 1
            runs[cache] = (rec, {k: v.detach().clone() for m in (G, D) for k, v in m.state_dict().items()})
        finally:
            O.WPREP_CACHE = True
    assert runs[True][0] == runs[False][0], (runs[True][0], runs[False][0])
    for k, v in runs[False][1].items():
        assert torch.equal(v, runs[True][1][k]), k
