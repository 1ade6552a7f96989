#!/usr/bin/env python3
"""Forward / dgrad / wgrad times of the generator's few-pixel layers (enc4-7, dec6-7 at B = 16, 256x256 input): direct kernels with
split-K.   python tools/small_layer_bench.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "terra-gan_amd"))
import torch
from tg_hip import ops as O
dev = torch.device("cuda:0")


def t(fn, reps=30):
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


g = torch.Generator().manual_seed(0)
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for name, hw, cin, cout, k, s, p in [("enc4", 32, 256, 512, 3, 2, 1), ("enc5", 16, 512, 512, 3, 2, 1), ("enc6", 8, 512, 512, 3, 2, 1),
                                     ("enc7", 4, 512, 512, 3, 2, 1), ("dec7", 4, 1024, 512, 3, 1, 1), ("dec6", 8, 1024, 512, 3, 1, 1)]:
    B = 16
    x = torch.randn(B, hw, hw, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, k, k, generator=g) * 0.02).contiguous(memory_format=torch.channels_last).to(dev)
    b = torch.zeros(cout).to(dev)
    m = (torch.rand(B, hw, hw, generator=g) > 0.2).float().to(dev)
    _, ratio = O.mask_update(m, k, s, p)
    y = O.conv_fwd(x, w, b, k, s, p, in_mask=m, ratio=ratio)
    dy = torch.randn(y.shape, generator=g).to(dev)
    gf = 2.0 * y.numel() * cin * k * k / 1e9
    row = {}
    row["fwd"] = t(lambda: O.conv_fwd(x, w, b, k, s, p, in_mask=m, ratio=ratio))
    row["dgrad"] = t(lambda: O.conv_dgrad(dy, w, tuple(x.shape), k, s, p, in_mask=m))
    row["wgrad"] = t(lambda: O.conv_wgrad(x, dy, w, k, s, p, in_mask=m if s == 2 else None, want_bias=False))
    print(name, " ".join(f"{op} {us:6.1f} us ({gf / us * 1e3:5.1f} TF)" for op, us in row.items()))
    for op, us in row.items():
        tot[op] += us
print("totals (us):", {k: round(v, 1) for k, v in tot.items()}, "sum", round(sum(tot.values()), 1))
