#!/usr/bin/env python3
"""Randomised check of the DIRECT convolution kernels (implicit-GEMM forward / dgrad incl. the merged parity classes of strided
dgrads, the weight-gradient kernel, the 1-channel-side kernels) against PyTorch-CPU fp64: random kernel sizes / strides / channel
counts / ragged spatial sizes, with the partial-conv mask and ratio.  Run with TG_NO_WINO=1 TG_NO_WINO22=1 TG_NO_S2D=1 to send the
stride-1 3x3 / 4x4 stride-2 / 5x5 stride-2 layers through them too.  Prints the worst error / tolerance; exits non-zero above 1.
    python tools/direct_fuzz.py [--cases 80] [--seed 0]"""
import argparse
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "terra-gan_amd"))
import torch
import torch.nn.functional as F
from tg_hip import ops as O


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def ratio_of(err_t, ref_t, rtol, atol):
    a, b = err_t.detach().double().cpu(), ref_t.detach().double().cpu()
    return float((a - b).abs().max()) / (atol + rtol * float(b.abs().max()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=80)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(args.seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    worst, worst_case = 0.0, None
    for case in range(args.cases):
        k, s = [(3, 1), (3, 2), (4, 2), (5, 2), (1, 1), (7, 2)][ri(0, 5)]
        p = {3: 1, 4: 1, 5: 2, 1: 0, 7: 3}[k]
        kind = ri(0, 9)
        if kind == 0:
            Cin, Cout = 1, 64 * ri(1, 2)                # 1-channel source
        elif kind == 1:
            Cin, Cout = 64, 1                           # 1-channel destination
            if k not in (3, 4):
                k, s, p = 3, 1, 1
        else:
            Cin, Cout = 4 * ri(1, 96), 4 * ri(1, 96)
            if ri(0, 5) == 0:
                Cin = ri(1, 9)                          # scalar-gather path
        B = ri(1, 4)
        H, W = ri(max(k, 6), 40), ri(max(k, 6), 48)
        if Cout == 1:
            W = 4 * ((W + 3) // 4)
        x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
        w = (torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / (k * Cin ** 0.5)).requires_grad_(True)
        b = torch.randn(Cout, generator=g, dtype=torch.float64, requires_grad=True)
        m = (torch.rand(B, 1, H, W, generator=g) > 0.3).double()
        ssum = F.conv2d(m, torch.ones(1, 1, k, k, dtype=torch.float64), None, s, p)
        ratio = (k * k) / (ssum + 1e-8) * (ssum > 0).double()
        z_ref = F.conv2d(x * m, w, b, s, p)
        y_ref = F.leaky_relu(z_ref * ratio, 0.2)
        gy = torch.randn(y_ref.shape, generator=g, dtype=torch.float64)
        xd, md = nhwc(x.detach().float()).to(dev), m[:, 0].float().contiguous().to(dev)
        wd = w.detach().float().contiguous(memory_format=torch.channels_last).to(dev)
        _mo, rd = O.mask_update(md, k, s, p)
        y = O.conv_fwd(xd, wd, b.detach().float().to(dev), k, s, p, in_mask=md, ratio=rd, act=O.ACT_LEAKY, slope=0.2)
        dyr = O.act_bwd(nhwc(gy.float()).to(dev), y, O.ACT_LEAKY, 0.2, ratio=rd, inplace=False)
        # the LeakyReLU gate is taken from the kernels' OWN forward output: an output within fp32 rounding of zero may have the other
        # sign in fp64, and one flipped gate moves every gradient by a whole term -- that is the activation's discontinuity, not
        # an error of the (linear) dgrad / wgrad kernels under test
        gate = torch.where(nchw(y).double().cpu() > 0, 1.0, 0.2)
        dz = gy * gate * ratio
        z_ref.backward(dz)
        dx = O.conv_dgrad(dyr, wd, tuple(xd.shape), k, s, p, in_mask=md)
        dw, db = O.conv_wgrad(xd, dyr, wd, k, s, p, in_mask=md)
        rs = {"fwd": ratio_of(nchw(y), y_ref, 2e-5, 1e-6), "act_bwd": ratio_of(nchw(dyr), dz, 1e-6, 1e-7),
              "dgrad": ratio_of(nchw(dx), x.grad, 1e-4, 1e-6), "wgrad": ratio_of(dw, w.grad, 1e-4, 1e-5),
              "bias": ratio_of(db, b.grad, 1e-4, 1e-5)}
        r = max(rs.values())
        if r > worst:
            worst, worst_case = r, (B, H, W, Cin, Cout, k, s, p, rs)
    print(f"{args.cases} cases, worst error / tolerance = {worst:.3f}  at {worst_case}")
    sys.exit(0 if worst <= 1.0 else 1)


if __name__ == "__main__":
    main()
