#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernels on the layer shapes of the 256x256 / B=16 train step.
Prints per-shape TFLOP/s measured with the library's own hipEvent hooks (tg_prof_*).
    python tools/conv_bench.py [--reps 5] [--only dec1,vgg1_2]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "terra-gan_amd"))
import torch  # noqa: E402

from tg_hip import lib as L  # noqa: E402
from tg_hip import ops as O  # noqa: E402

# name, B, H, W, Cin, Cout, k, s, p, masked
SHAPES = [
    ("vgg1_2", 32, 256, 256, 64, 64, 3, 1, 1, False),
    ("dec1", 16, 256, 256, 64, 64, 3, 1, 1, True),
    ("dec2", 16, 128, 128, 192, 64, 3, 1, 1, True),
    ("dec3", 16, 64, 64, 384, 128, 3, 1, 1, True),
    ("dec4", 16, 32, 32, 768, 256, 3, 1, 1, True),
    ("dec5", 16, 16, 16, 1024, 512, 3, 1, 1, True),
    ("vgg2_2", 32, 128, 128, 128, 128, 3, 1, 1, False),
    ("vgg3_2", 32, 64, 64, 256, 256, 3, 1, 1, False),
    ("enc2", 16, 128, 128, 64, 128, 5, 2, 2, True),
    ("enc3", 16, 64, 64, 128, 256, 5, 2, 2, True),
    ("enc5", 16, 16, 16, 512, 512, 3, 2, 1, True),
    ("d1", 16, 128, 128, 64, 128, 4, 2, 1, False),
    ("d3", 16, 32, 32, 256, 512, 4, 2, 1, False),
    ("final", 16, 256, 256, 64, 1, 3, 1, 1, False),
    ("enc1", 16, 256, 256, 1, 64, 7, 2, 3, True),
    ("d0", 32, 256, 256, 1, 64, 4, 2, 1, False),
]


def summary(lib, kind):
    ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    lib.tg_prof_summary(kind, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by))
    return ms.value, n.value, fl.value, by.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--wino4", action="store_true", help="request Winograd F(4x4,3x3) for the unmasked stride-1 3x3 layers (the VGG trunk)")
    ap.add_argument("--wgrad-mask", action="store_true", help="pass the input mask to stride-1 wgrads too (direct kernel)")
    args = ap.parse_args()
    lib = L.load()
    O.set_precision(args.precision)
    dev = torch.device("cuda:0")
    only = set(filter(None, args.only.split(",")))
    print(f"{'layer':8s} {'op':6s} {'ms':>8s} {'TF':>7s} {'algGB/s':>8s}")
    for name, B, H, W, Cin, Cout, k, s, p, masked in SHAPES:
        if only and name not in only:
            continue
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, H, W, Cin, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).contiguous(memory_format=torch.channels_last).to(dev)
        bias = torch.randn(Cout, generator=g).to(dev)
        m = (torch.rand(B, H, W, generator=g) > 0.2).float().to(dev) if masked else None
        ratio = None
        if masked:
            _, ratio = O.mask_update(m, k, s, p)
        y = O.conv_fwd(x, w, bias, k, s, p, in_mask=m, ratio=ratio)
        dy = torch.randn(y.shape, generator=g).to(dev)
        w4 = args.wino4 and not masked
        for op in ("fwd", "dgrad", "wgrad"):
            fn = {"fwd": lambda: O.conv_fwd(x, w, bias, k, s, p, in_mask=m, ratio=ratio, wino4=w4),
                  "dgrad": lambda: O.conv_dgrad(dy, w, tuple(x.shape), k, s, p, in_mask=m, wino4=w4),
                  # the engine hands stride-1 decoder layers a pre-masked concat tensor: no in_mask for their wgrad
                  "wgrad": lambda: O.conv_wgrad(x, dy, w, k, s, p, in_mask=None if (s == 1 and not args.wgrad_mask) else m)}[op]
            fn()
            torch.cuda.synchronize()
            lib.tg_prof_enable(1)
            for _ in range(args.reps):
                fn()
            torch.cuda.synchronize()
            lib.tg_prof_enable(0)
            ms0, n0, fl0, by0 = summary(lib, 0)
            ms1, n1, fl1, by1 = summary(lib, 1)
            ms2, n2, fl2, by2 = summary(lib, 2)
            ms3, n3, fl3, by3 = summary(lib, 3)
            ms, fl, by = ms0 + ms1 + ms2 + ms3, fl0 + fl1 + fl2 + fl3, by0 + by1 + by2 + by3
            print(f"{name:8s} {op:6s} {ms / args.reps:8.3f} {fl / ms / 1e9 if ms else 0:7.1f} {by / ms / 1e6 if ms else 0:8.1f}", flush=True)


if __name__ == "__main__":
    main()
