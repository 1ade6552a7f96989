#!/usr/bin/env python3
"""Where the Winograd weight-gradient kernels spend their time: wino_wgrad_kernel (F(3x3,2x2), stride-1 3x3 layers) and
wino22_wgrad_kernel (F(2x2,2x2), the discriminator's 4x4 stride-2 layers).  A launch is  splits x (Cout/64) x (Cin/64)  work
items; each walks its share of the strips (8 tiles per K step) and writes a slab that the reduce kernel sums.  The batch size is
swept at fixed channels: time = fixed part (prologue, slab stores, reduce) + strips x per-strip time; a least-squares line
gives both.   python tools/wgrad_item_overhead.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "terra-gan_amd"))
import torch  # noqa: E402
from tg_hip import ops as O  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def fit(xs, ys):
    n = len(xs)
    mx, my = sum(xs) / n, sum(ys) / n
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    return my - b * mx, b


g = torch.Generator().manual_seed(0)
for name, k, s, p, Cin, Cout, hw, batches in [("wino_wgrad  (dec1-shaped 64->64, 256^2)", 3, 1, 1, 64, 64, 256, (4, 8, 16, 32)),
                                              ("wino_wgrad  (dec3-shaped 384->128, 64^2)", 3, 1, 1, 384, 128, 64, (8, 16, 32, 64)),
                                              ("wino22_wgrad (d2-shaped 64->128, 128^2 in)", 4, 2, 1, 64, 128, 128, (8, 16, 32, 64)),
                                              ("wino22_wgrad (d8-shaped 256->512, 32^2 in)", 4, 2, 1, 256, 512, 32, (16, 32, 64, 128))]:
    xs, ys = [], []
    for B in batches:
        x = torch.randn(B, hw, hw, Cin, generator=g).to(dev)
        ho = (hw + 2 * p - k) // s + 1
        dy = torch.randn(B, ho, ho, Cout, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).contiguous(memory_format=torch.channels_last).to(dev)
        us = t(lambda: O.conv_wgrad(x, dy, w, k, s, p))
        gf = 2.0 * B * ho * ho * Cout * Cin * k * k / 1e9
        xs.append(B)
        ys.append(us)
        print(f"  {name}  B {B:4d}: {us:8.1f} us  {gf / us * 1e-3:7.1f} algorithmic TF")
    a, b = fit(xs, ys)
    print(f"{name}: fixed part {a:6.1f} us, {b:6.2f} us per image of the batch -> at the train step's batch the fixed part is "
          f"{a / (a + b * batches[-2]) * 100:4.1f} % of the launch")
