#!/usr/bin/env python3
"""conv -> ReLU -> 2x2 max-pool of the VGG trunk's pooled layers (conv1_2, conv2_2 over 2B = 32 images): one tg_conv_fwd_pool call
against tg_conv_fwd_p + tg_maxpool2_fwd.   python tools/pool_fuse_bench.py"""
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
for name, B, hw, cin, cout in [("conv1_2", 32, 256, 64, 64), ("conv2_2", 32, 128, 128, 128)]:
    x = torch.randn(B, hw, hw, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.03).contiguous(memory_format=torch.channels_last).to(dev)
    b = torch.zeros(cout).to(dev)
    y = O.conv_fwd(x, w, b, 3, 1, 1, act=O.ACT_RELU)
    for rnd in range(3):          # interleaved rounds: the first one also warms the clocks
        conv = t(lambda: O.conv_fwd(x, w, b, 3, 1, 1, act=O.ACT_RELU))
        pool = t(lambda: O.maxpool2_fwd(y))
        both = t(lambda: O.maxpool2_fwd(O.conv_fwd(x, w, b, 3, 1, 1, act=O.ACT_RELU)))
        fused = t(lambda: O.conv_fwd(x, w, b, 3, 1, 1, act=O.ACT_RELU, pool=True))
        print(f"{name} [{rnd}]: conv {conv:7.1f} us, pool {pool:6.1f} us, conv + pool {both:7.1f} us, fused {fused:7.1f} us")
