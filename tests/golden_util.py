"""Helpers shared by the golden-fixture tests (oracle on CPU, HIP path on GPU)."""
import os

import numpy as np
import torch

from tests.golden.make_golden import MASK_KINDS, PCONV_CASES, mask_case  # noqa: F401  (input recipes)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))
    return _cache[name]


def check(gold, key, t, atol=2e-6, rtol=1e-5, scale_by_max=False):
    """Compare tensor `t` with fixture entry `key` (full tensor, or sum/abssum/l2/sample summary).
    scale_by_max: gradient rule of SURVEY §8c -- max|d| <= rtol*max|g| + atol."""
    t = t.detach().double().flatten().cpu()
    if key + "/full" in gold:
        ref = torch.from_numpy(gold[key + "/full"]).double()
        assert ref.numel() == t.numel(), f"{key}: numel {t.numel()} vs {ref.numel()}"
        _cmp(key, t, ref, atol, rtol, scale_by_max)
        return
    stride = int(gold[key + "/stride"])
    ref = torch.from_numpy(gold[key + "/sample"]).double()
    _cmp(key + "[sample]", t[::stride][:512], ref, atol, rtol, scale_by_max)
    n = t.numel()
    l2 = float(gold[key + "/l2"])
    # aggregate checks: error of a sum of n rounded terms
    tol = rtol * float(gold[key + "/abssum"]) + atol * n
    assert abs(float(t.sum()) - float(gold[key + "/sum"])) <= tol, f"{key}: sum"
    assert abs(float(t.norm()) - l2) <= rtol * l2 + atol * n ** 0.5, f"{key}: l2"


def _cmp(key, t, ref, atol, rtol, scale_by_max):
    err = (t - ref).abs()
    if scale_by_max:
        bound = rtol * float(ref.abs().max()) + atol
        assert float(err.max()) <= bound, f"{key}: max err {float(err.max()):.3e} > {bound:.3e}"
    else:
        bad = err > (atol + rtol * ref.abs())
        assert not bool(bad.any()), (f"{key}: {int(bad.sum())}/{t.numel()} off, max err "
                                     f"{float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e})")


# --------------------------------------------------------------------------------------------------
# stated tolerance + the reference arithmetic's own fp32 uncertainty (tests/golden/steps_unc.npz)
# --------------------------------------------------------------------------------------------------
K_UNC = 5.0          # how many fp32-vs-fp64 deviations of the oracle a result may sit away from the reference
K_DRIFT = 20.0       # same, for quantities AFTER an Adam update (steps >= 1): two fp32 trajectories separate chaotically
_ratios = {}
_failures = []
_tensors = {}        # per tensor / quantity: err, the STATED bound alone and the widened bound the test asserts


def begin():
    _ratios.clear()
    _failures.clear()
    _tensors.clear()


def expect(ok, msg):
    """Deferred assertion: a parity test reports ALL its violations (and still writes its ratio table) before failing."""
    if not ok:
        _failures.append(msg)


def finish(name):
    dump_ratios(name)
    assert not _failures, f"{len(_failures)} parity violations:\n  " + "\n  ".join(str(f) for f in _failures[:40])


def record(group, key, ratio, err=None, bound=None, stated=None):
    """Keep the worst err/bound ratio per group; tests dump them with dump_ratios() (evidence for the parity report).
    `stated`: the stated tolerance of SURVEY 8c ALONE (no uncertainty term) -- kept per tensor next to the asserted bound, and
    counted per group (n_within_stated), so the record says how much of the pass rests on the widening."""
    g = _ratios.setdefault(group, {"worst_ratio": 0.0, "worst_key": None, "n": 0, "n_stated": 0, "n_within_stated": 0,
                                   "worst_stated_ratio": 0.0, "worst_stated_key": None})
    g["n"] += 1
    if ratio >= g["worst_ratio"]:
        g.update(worst_ratio=float(ratio), worst_key=key, err=None if err is None else float(err),
                 bound=None if bound is None else float(bound))
    if stated is not None and err is not None:
        sr = float(err) / float(stated) if stated > 0 else float("inf")
        g["n_stated"] += 1
        g["n_within_stated"] += int(sr <= 1.0)
        if sr >= g["worst_stated_ratio"]:
            g.update(worst_stated_ratio=sr, worst_stated_key=key)
        _tensors[f"{group}:{key}"] = {"err": float(err), "stated_bound": float(stated), "asserted_bound": float(bound),
                                      "ratio_stated": sr, "ratio_asserted": float(ratio)}


def dump_ratios(name):
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out", "parity")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(_ratios, f, indent=1, sort_keys=True)
        if _tensors:
            with open(os.path.join(out, name + "_per_tensor.json"), "w") as f:
                json.dump(_tensors, f, indent=0, sort_keys=True)
    except OSError:
        pass
    print("\nparity ratios (err / bound, worst per group):")
    for g, v in sorted(_ratios.items()):
        st = f"; within the STATED bound alone: {v['n_within_stated']}/{v['n_stated']}, worst {v['worst_stated_ratio']:.2f} ({v['worst_stated_key']})" \
            if v.get("n_stated") else ""
        print(f"  {g:32s} {v['worst_ratio']:.3f}  ({v['worst_key']}, err {v.get('err')}, bound {v.get('bound')}, n={v['n']}){st}")


def check_unc(gold, unc, key, t, group, rel=1e-3, atol=1e-10, k=None):
    """Gradient rule of SURVEY 8c, max|d| <= rel*max|g|, plus k x the oracle's fp32-vs-fp64 deviation of the SAME tensor
    (unc[key] = [max|d|, sum|d|, ||d||_2, max|g|]).  Returns the worst err/bound ratio (asserted <= 1)."""
    k = K_UNC if k is None else k
    t = t.detach().double().flatten().cpu()
    umax, usum, ul2, gmax = [float(v) for v in unc[key]]
    bound = rel * gmax + k * umax + atol
    checks = []
    if key + "/full" in gold:
        ref = torch.from_numpy(gold[key + "/full"]).double()
        assert ref.numel() == t.numel(), f"{key}: numel {t.numel()} vs {ref.numel()}"
        checks.append(("max", float((t - ref).abs().max()), bound, rel * gmax + atol))
    else:
        stride = int(gold[key + "/stride"])
        ref = torch.from_numpy(gold[key + "/sample"]).double()
        n = t.numel()
        checks.append(("sample max", float((t[::stride][:512] - ref).abs().max()), bound, rel * gmax + atol))
        checks.append(("sum", abs(float(t.sum()) - float(gold[key + "/sum"])),
                       rel * float(gold[key + "/abssum"]) + k * usum + atol * n, rel * float(gold[key + "/abssum"]) + atol * n))
        l2 = float(gold[key + "/l2"])
        checks.append(("l2", abs(float(t.norm()) - l2), rel * l2 + k * ul2 + atol * n ** 0.5, rel * l2 + atol * n ** 0.5))
    worst = 0.0
    for what, err, bnd, stated in checks:
        record(group, f"{key} [{what}]", err / bnd, err, bnd, stated=stated)
        expect(err <= bnd, f"{key} [{what}]: err {err:.3e} > bound {bnd:.3e} (oracle fp32-vs-fp64 max dev {umax:.3e}, max|g| {gmax:.3e})")
        worst = max(worst, err / bnd)
    return worst
