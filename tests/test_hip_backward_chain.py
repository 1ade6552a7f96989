"""The generator's backward chain at the headline size (B = 16, 256x256: BASELINE configs[1]), layer by layer, against the
fp64 oracle fixtures (tests/golden/steps_chain*.npz, make_golden.py steps_chain / steps_chain_seeds / steps_chain_gy) --
the round-3 verdict's first item: where does the HIP backward leave the spread of the reference arithmetic's own fp32
evaluations?

Per tensor the fixture holds a strided sample of the fp64 value and the deviation of THREE CPU fp32 evaluations (oneDNN
convolutions = what the reference runs, ATen's native convolutions, one thread) from it on the same sample; the test
measures the HIP path's deviation on that sample and reports  ratio = HIP deviation / largest CPU fp32 deviation  (rms and
max), walking from `final` back to `enc1`.  What was found (profiles/r04_backward_chain_c2.json):

  * with a SMOOTH upstream gradient (sum(gen * gy), the same forward, the same backward kernels) the chain sits inside the
    CPU spread: geometric mean ratio 0.3 over five data seeds, 95 % of the tensors within 2x;
  * driven by the train step's own loss, the ratios depend on the draw: the loss stack's gradient is DISCONTINUOUS in the
    generated image (|p - t| of the L1 / boundary / perceptual terms, the ReLU / LeakyReLU / max-pool gates of the VGG trunk
    and the discriminator) -- where p ~ t to rounding, one boundary-band pixel flips a gradient of 0.5 / sum(band) = 27 % of
    max|dL/dgen|, and any two fp32 evaluations (the CPU ones among themselves too) may disagree on it; the generator's
    backward then carries that localised difference into every gradient tensor.  The tests below therefore hold the smooth
    chain to the CPU spread and RECORD the train-step table."""
import json
import math
import os

import pytest
import torch

from tests import chain_util as CU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tg_hip import lib
    lib.load()
    return torch.device("cuda:0")


def _stats(rows):
    sel = [(k, r) for k, r in rows.items() if k.startswith(("bwd", "grad")) and r["ratio_rms"] > 0]
    rr = [r["ratio_rms"] for _k, r in sel]
    gm = math.exp(sum(math.log(x) for x in rr) / len(rr))
    worst = max(sel, key=lambda kr: kr[1]["ratio_rms"])
    return {"n": len(rr), "geo_mean_ratio_rms": gm, "frac_above_2": sum(x > 2 for x in rr) / len(rr),
            "max_ratio_rms": max(rr), "worst": worst[0]}


def _write(name, rows, stats):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out", "parity")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name + ".json"), "w") as f:
        json.dump({"stats": stats, "order": CU.ordered_keys(rows),
                   "rows": {k: {q: r[q] for q in ("ratio_rms", "ratio_max", "hip_rms", "cpu_rms", "hip_max", "cpu_max", "ref_max", "stated")}
                            for k, r in rows.items()}}, f, indent=1)


@pytest.mark.parametrize("seed", [500, 501, 502, 503, 504])
def test_generator_backward_chain_smooth_upstream(dev, seed):
    """Same forward, same backward kernels, a seeded smooth upstream gradient: every layer's activation gradient and every
    parameter gradient against fp64.  Held to the spread of the CPU fp32 evaluations: geometric mean of the rms ratios <= 1.5,
    at most 30 % of the tensors beyond 2x, none beyond 10x (measured over the five seeds: 0.08 ... 1.06, 0 ... 23 %, <= 5.4)."""
    rows = CU.measure_chain(dev, f"c2_b16_256_gy_s{seed}", fixture="steps_chain_gy")
    st = _stats(rows)
    print(f"\nsmooth-upstream backward chain, seed {seed}: {st}")
    print(CU.format_table(rows))
    _write(f"backward_chain_gy_s{seed}", rows, st)
    assert st["geo_mean_ratio_rms"] <= 1.5 and st["frac_above_2"] <= 0.30 and st["max_ratio_rms"] <= 10.0, st


def test_train_step_backward_chain_table(dev):
    """The walk the verdict asked for, on the train step itself (the reference fixture's batch: data seed 500): recorded to
    gpurun_out/parity/backward_chain_c2.json (committed as profiles/r04_backward_chain_c2.json together with four more seeds
    and the direct-kernel / three-launch-BatchNorm A/B runs).  Asserted: what the loss stack hands the generator (bwd/gen) and
    the forward activations are inside twice the CPU spread; the chain as a whole within the range the five recorded draws
    span (geometric mean 0.3 ... 2.3)."""
    rows = CU.measure_chain(dev, "c2_b16_256")
    st = _stats(rows)
    print(f"\ntrain-step backward chain: {st}")
    print(CU.format_table(rows))
    _write("backward_chain_c2", rows, st)
    assert rows["bwd/gen"]["ratio_rms"] <= 2.5, rows["bwd/gen"]
    for k, r in rows.items():
        if k.startswith("fwd/dec") or k in ("fwd/final", "fwd/gen"):
            assert r["ratio_rms"] <= 2.0, (k, r["ratio_rms"])
    assert st["geo_mean_ratio_rms"] <= 4.0 and st["max_ratio_rms"] <= 25.0, st
