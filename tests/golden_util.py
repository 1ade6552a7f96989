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
