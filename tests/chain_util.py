"""The generator's backward chain at the headline size, layer by layer: HIP against the fp64 oracle fixture
(tests/golden/steps_chain.npz) next to the spread of three CPU fp32 evaluations of the same arithmetic.  Shared by
tests/test_hip_backward_chain.py and tools/backward_chain.py."""
import numpy as np
import torch

from tests import golden_util as GU

# backward order: what the loss hands the generator, then final, dec1 ... dec7, enc7 ... enc1
CHAIN = ["gen", "final"] + [f"dec{i}" for i in range(1, 8)] + [f"enc{i}" for i in range(7, 0, -1)]


def layer_params(name):
    if name == "gen":
        return []
    if name == "final":
        return ["final.weight", "final.bias"]
    return [f"{name}.input_conv.weight", f"{name}.input_conv.bias", f"{name}.bn.weight", f"{name}.bn.bias"]


def measure_chain(dev, tag="c2_b16_256", seed=0, fixture="steps_chain"):
    """One generator forward + backward of the train step on the HIP path with engine probes -> rows
    {key: dict(hip_max, hip_rms, cpu_max, cpu_rms, ratio_max, ratio_rms, ref_max, ref_rms, stated)} where
    *_max / *_rms are the deviation from the fp64 sample, cpu_* the LARGEST over the three CPU fp32 evaluations and
    stated = hip_max / (1e-3 * max|ref|) (SURVEY 8c's gradient rule; for fwd rows: hip_max / 2e-6)."""
    from mvp_gan.src.models import Discriminator, PConvUNet
    from mvp_gan.src.train import train_step
    from mvp_gan.src.utils.losses import InpaintingLoss
    from oracle import terragan_oracle as Orc
    from tg_hip import engine as E
    gold = GU.load(fixture)
    cfg = [int(v) for v in gold[f"{tag}/cfg"]]
    b, size, seed0, _ns = cfg[:4]
    gy_mode = len(cfg) > 4 and cfg[4] == 1        # a seeded smooth upstream gradient instead of the loss stack's (make_golden.py)
    torch.manual_seed(seed)
    G, D = PConvUNet(), Discriminator()
    crit = InpaintingLoss(0.1, 0.1, device=torch.device("cpu"))
    G, D, crit = G.to(dev), D.to(dev), crit.to(dev)
    oG, oD = torch.optim.Adam(G.parameters(), lr=2e-4), torch.optim.Adam(D.parameters(), lr=2e-4)
    real, mask = Orc.synth_batch(b, size, seed0)
    got = {}

    def probe(kind, name, t):
        key = f"{tag}/{kind}/{name}"
        if key + "/ref" in gold and key not in got:          # the first call per key: the generator step (not validation etc.)
            stride = int(gold[key + "/stride"])
            n = gold[key + "/ref"].shape[0]
            got[key] = t.detach().reshape(-1)[::stride][:n].double().cpu()

    E.PROBE = probe
    try:
        if gy_mode:
            from tg_hip import ops as O
            GP = G._tensors()
            rb, mb = real.to(dev).reshape(b, size, size), mask.to(dev).reshape(b, size, size)
            gen, gctx = E.generator_forward(GP, O.mul(rb, mb), mb, True)
            gy = (torch.randn(b, 1, size, size, generator=torch.Generator().manual_seed(5)) / (b * size * size)).to(dev)
            grads, _ = E.generator_backward(GP, gctx, gy.reshape(b, size, size).contiguous())
        else:
            train_step(G, D, crit, oG, oD, real.to(dev), mask.to(dev))
            grads = {k: p_.grad for k, p_ in G.named_parameters() if p_.requires_grad}
        torch.cuda.synchronize()
    finally:
        E.PROBE = None
    for k, g_ in grads.items():
        key = f"{tag}/grad/{k}"
        if key + "/ref" in gold:
            stride = int(gold[key + "/stride"])
            n = gold[key + "/ref"].shape[0]
            got[key] = g_.detach().flatten()[::stride][:n].double().cpu()           # logical OIHW order
    rows = {}
    for key, mine in got.items():
        ref = torch.from_numpy(gold[key + "/ref"]).double()
        assert ref.shape == mine.shape, (key, ref.shape, mine.shape)
        d = mine - ref
        dev_ = gold[key + "/dev"]                           # [variant][max, rms]
        ref_max, ref_rms, _n = [float(v) for v in gold[key + "/scale"]]
        hip_max, hip_rms = float(d.abs().max()), float(d.pow(2).mean().sqrt())
        cpu_max, cpu_rms = float(dev_[:, 0].max()), float(dev_[:, 1].max())
        kind = key.split("/")[1]
        stated = hip_max / (2e-6 if kind == "fwd" else 1e-3 * ref_max + 1e-30)
        rows[key[len(tag) + 1:]] = dict(hip_max=hip_max, hip_rms=hip_rms, cpu_max=cpu_max, cpu_rms=cpu_rms,
                                        ratio_max=hip_max / (cpu_max + 1e-300), ratio_rms=hip_rms / (cpu_rms + 1e-300),
                                        ref_max=ref_max, ref_rms=ref_rms, stated=stated,
                                        cpu_variants=[[float(v) for v in r] for r in dev_])
    return rows


def ordered_keys(rows):
    out = []
    for name in CHAIN:
        for k in [f"fwd/{name}", f"bwd/{name}"] + [f"grad/{p}" for p in layer_params(name)]:
            if k in rows:
                out.append(k)
    return out


def format_table(rows):
    lines = [f"{'tensor':34s} {'max|ref|':>10s} {'hip max':>10s} {'cpu max':>10s} {'r_max':>7s} {'hip rms':>10s} {'cpu rms':>10s} {'r_rms':>7s} {'stated':>7s}"]
    for k in ordered_keys(rows):
        r = rows[k]
        lines.append(f"{k:34s} {r['ref_max']:10.3e} {r['hip_max']:10.3e} {r['cpu_max']:10.3e} {r['ratio_max']:7.2f} "
                     f"{r['hip_rms']:10.3e} {r['cpu_rms']:10.3e} {r['ratio_rms']:7.2f} {r['stated']:7.2f}")
    return "\n".join(lines)
