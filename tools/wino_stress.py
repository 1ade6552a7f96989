#!/usr/bin/env python3
"""Randomised shape sweep of the Winograd / space-to-depth conv paths against PyTorch-CPU fp64 (run on an MI355X).
    python tools/wino_stress.py [--n 40] [--seed 0] [--big]
--big: batches of 6-12 and up to 112 pixels a side, i.e. several work items per persistent workgroup (cross-item staging
pipeline, interleaved walk), and every fourth case a 4x4 / stride-2 / pad-1 layer (the F(2x2,2x2) kernels)."""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "terra-gan_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tg_hip import lib as L  # noqa: E402
from tg_hip import ops as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--big", action="store_true")
    args = ap.parse_args()
    L.load()
    dev = torch.device("cuda:0")
    rnd = random.Random(args.seed)
    worst = 0.0
    for it in range(args.n):
        s2d = it % 4 == 3
        B = rnd.choice([6, 8, 9, 12]) if args.big else rnd.choice([1, 2, 3, 5])
        if args.big and s2d:
            H, W = 2 * rnd.randint(24, 56), 2 * rnd.randint(24, 56)
            Cin, Cout, k, s, p = rnd.choice([64, 128]), rnd.choice([64, 128]), 4, 2, 1
        elif args.big:
            H, W = rnd.randint(48, 112), rnd.randint(48, 112)
            Cin, Cout, k, s, p = rnd.choice([16, 64, 64, 128]), rnd.choice([64, 128]), 3, 1, rnd.choice([0, 1, 1])
        elif s2d:
            H, W = 2 * rnd.randint(32, 48), 2 * rnd.randint(32, 56)
            Cin, Cout, k, s, p = rnd.choice([16, 32, 64, 80]), rnd.choice([64, 128, 192]), 5, 2, 2
        else:
            H, W = rnd.randint(16, 70), rnd.randint(16, 70)
            Cin, Cout, k, s, p = rnd.choice([8, 24, 64, 72, 128, 136]), rnd.choice([64, 128, 192]), 3, 1, rnd.choice([0, 1, 1, 2])
        if H + 2 * p - k < 15 or W + 2 * p - k < 15:
            continue
        g = torch.Generator().manual_seed(1000 + it)
        x = torch.randn(B, H, W, Cin, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / (k * Cin ** 0.5)
        bias = torch.randn(Cout, generator=g) * 0.1
        mask = (torch.rand(B, H, W, generator=g) > 0.3).float()
        use_mask = rnd.random() < (0.3 if args.big else 0.6)
        wd = w.contiguous(memory_format=torch.channels_last).to(dev)
        m_d = mask.to(dev) if use_mask else None
        xm = (x * mask[..., None] if use_mask else x).permute(0, 3, 1, 2).double()
        ref = F.conv2d(xm, w.double(), bias.double(), s, p).permute(0, 2, 3, 1)
        y = O.conv_fwd(x.to(dev), wd, bias.to(dev), k, s, p, in_mask=m_d)
        e_f = (y.cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        dy = torch.randn(ref.shape, generator=g)
        refdx = torch.autograd.grad(F.conv2d(xm.requires_grad_(True), w.double(), None, s, p), xm, dy.permute(0, 3, 1, 2).double())[0]
        refdx = refdx.permute(0, 2, 3, 1) * (mask[..., None].double() if use_mask else 1.0)
        dx = O.conv_dgrad(dy.to(dev), wd, (B, H, W, Cin), k, s, p, in_mask=m_d)
        e_d = (dx.cpu().double() - refdx).abs().max().item() / max(1.0, refdx.abs().max().item())
        refdw = torch.nn.grad.conv2d_weight(xm.detach(), (Cout, Cin, k, k), dy.permute(0, 3, 1, 2).double(), stride=s, padding=p)
        dw, _ = O.conv_wgrad(x.to(dev), dy.to(dev), wd, k, s, p, in_mask=m_d)
        e_w = (dw.cpu().double() - refdw).abs().max().item() / refdw.abs().max().item()
        worst = max(worst, e_f, e_d, e_w)
        flag = "" if max(e_f, e_d, e_w) < 2e-5 else "   <-- LARGE"
        print(f"{it:3d} B{B} {H}x{W} {Cin}->{Cout} k{k}s{s}p{p} mask={int(use_mask)}  fwd {e_f:.1e} dgrad {e_d:.1e} wgrad {e_w:.1e}{flag}", flush=True)
    print("worst relative error", worst)
    assert worst < 2e-5


if __name__ == "__main__":
    main()
