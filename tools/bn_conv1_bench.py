#!/usr/bin/env python3
"""dec1's BatchNorm backward at the headline size (B = 16, 256 x 256 x 64): conv_dgrad(final) + bn_act_bwd against bn_act_bwd_conv1
(final's input gradient recomputed in both passes).   python tools/bn_conv1_bench.py   [TG_BN_CONV1_GRID=n]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "terra-gan_amd"))
import torch
from tg_hip import ops as O
dev = torch.device("cuda:0")


def t(fn, reps=20):
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


g = torch.Generator().manual_seed(0)
B, H, W, C = 16, 256, 256, 64
y = torch.randn(B, H, W, C, generator=g).to(dev)
gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
ratio = torch.rand(B, H, W, generator=g).to(dev)
w = (torch.randn(1, C, 3, 3, generator=g) * 0.1).contiguous(memory_format=torch.channels_last).to(dev)
dz = torch.randn(B, H, W, 1, generator=g).to(dev)
mean, rstd = O.bn_stats(y)
for rnd in range(3):
    two = t(lambda: O.bn_act_bwd(O.conv_dgrad(dz, w, tuple(y.shape), 3, 1, 1), y, mean, rstd, gamma, beta, O.ACT_RELU, ratio=ratio))
    one = t(lambda: O.bn_act_bwd_conv1(dz, w, y, mean, rstd, gamma, beta, O.ACT_RELU, ratio=ratio))
    print(f"[{rnd}] dgrad + bn_act_bwd {two:7.1f} us   bn_act_bwd_conv1 {one:7.1f} us")
