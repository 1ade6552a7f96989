#!/usr/bin/env python3
"""Time the element-wise / reduction passes of the train step at the layer shapes of BASELINE configs[1] (256x256, B=16):
BatchNorm forward (statistics + apply), BatchNorm backward (sums + apply), upsample-concat forward / backward, 2x2 max-pool.
Prints us per call and the achieved GB/s on the algorithmic bytes.   python tools/ew_bench.py [--reps 30]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "terra-gan_amd"))
import torch  # noqa: E402
from tg_hip import ops as O  # noqa: E402


def t(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 30
    dev = torch.device("cuda:0")
    B = 16
    tot = {}
    for name, hw, C in [("dec1", 256, 64), ("dec2/enc1", 128, 64), ("dec3/enc2", 64, 128), ("dec4/enc3", 32, 256), ("dec5/enc4", 16, 512)]:
        rows = B * hw * hw
        y = torch.randn(B, hw, hw, C, device=dev)
        g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
        rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.long, device=dev)
        ratio = torch.rand(B, hw, hw, device=dev) + 0.5
        out = torch.empty_like(y)
        mean, rstd, _ = O.bn_fwd(y, g, b, O.ACT_RELU, 0.0, rm, rv, nbt, out=out)
        da = torch.randn_like(y)
        nb = y.numel() * 4
        us = t(lambda: O.bn_stats(y, rm, rv, nbt), reps)
        us2 = t(lambda: O.bn_act_fwd(y, mean, rstd, g, b, O.ACT_RELU, out=out), reps)
        us3 = t(lambda: O.bn_act_bwd(da, y, mean, rstd, g, b, O.ACT_RELU, ratio=ratio, inplace=True), reps)
        print(f"{name:10s} rows {rows:8d} C {C:4d}  bn_stats {us:7.1f} us ({nb / us / 1e3:6.0f} GB/s)  bn_act_fwd {us2:7.1f} us ({2 * nb / us2 / 1e3:6.0f} GB/s)"
              f"  bn_act_bwd {us3:7.1f} us ({5 * nb / us3 / 1e3:6.0f} GB/s on 5 tensor passes)")
        for k, v in (("bn_stats", us), ("bn_act_fwd", us2), ("bn_act_bwd", us3)):
            tot[k] = tot.get(k, 0) + v * (2 if hw < 256 else 1)          # encoder + decoder layer of that shape
    for name, h, Cu, Cs in [("dec1", 128, 64, 0), ("dec2", 64, 128, 64), ("dec3", 32, 256, 128), ("dec4", 16, 512, 256), ("dec5", 8, 512, 512)]:
        up = torch.randn(B, h, h, Cu, device=dev)
        skip = torch.randn(B, 2 * h, 2 * h, Cs, device=dev) if Cs else None
        m = (torch.rand(B, 2 * h, 2 * h, device=dev) > 0.2).float()
        cat = O.upcat_fwd(up, skip, 2 * h, 2 * h, out_mask=m)
        us = t(lambda: O.upcat_fwd(up, skip, 2 * h, 2 * h, out_mask=m), reps)
        dcat = torch.randn_like(cat)
        us2 = t(lambda: O.upcat_bwd(dcat, h, h, Cu), reps)
        nbf = (up.numel() + (skip.numel() if Cs else 0) + cat.numel()) * 4
        print(f"{name:10s} upcat_fwd {us:7.1f} us ({nbf / us / 1e3:6.0f} GB/s)  upcat_bwd {us2:7.1f} us ({nbf / us2 / 1e3:6.0f} GB/s)")
        tot["upcat_fwd"] = tot.get("upcat_fwd", 0) + us
        tot["upcat_bwd"] = tot.get("upcat_bwd", 0) + us2
    for name, n, hw, C in [("pool1", 32, 256, 64), ("pool2", 32, 128, 128)]:
        x = torch.randn(n, hw, hw, C, device=dev).relu_()
        us = t(lambda: O.maxpool2_fwd(x), reps)
        d = torch.randn(n // 2, hw // 2, hw // 2, C, device=dev)
        xh = x[: n // 2].contiguous()
        us2 = t(lambda: O.maxpool2_bwd(d, xh, relu_gate=True), reps)
        print(f"{name:10s} maxpool_fwd {us:7.1f} us ({x.numel() * 5 / us / 1e3:6.0f} GB/s)  maxpool_bwd {us2:7.1f} us ({xh.numel() * 9 / us2 / 1e3:6.0f} GB/s)")
        tot["maxpool_fwd"] = tot.get("maxpool_fwd", 0) + us
        tot["maxpool_bwd"] = tot.get("maxpool_bwd", 0) + us2
    print("per-step totals (us):", {k: round(v, 1) for k, v in tot.items()}, "sum", round(sum(tot.values()), 1))


if __name__ == "__main__":
    main()
