#!/usr/bin/env python3
"""Dump every generator-layer activation / activation gradient of one train step (engine probes) in a compact form, to compare
two library configurations offline (one process per configuration: the TG_* switches are read once):
small tensors in full, big ones as a strided sample plus a channel-summed map over 8x8-pixel blocks (where two runs differ).

    python tools/chain_dump.py --seed 504 --out gpurun_out/dump_default.npz
    TG_NO_BN_SMALL=1 python tools/chain_dump.py --seed 504 --out gpurun_out/dump_nobnsmall.npz
    python tools/chain_dump.py --compare gpurun_out/dump_default.npz gpurun_out/dump_nobnsmall.npz
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "terra-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("TERRAGAN_ALLOW_STANDIN_VGG", "1")


def dump(args):
    import torch
    from mvp_gan.src.models import Discriminator, PConvUNet
    from mvp_gan.src.train import train_step
    from mvp_gan.src.utils.losses import InpaintingLoss
    from oracle import terragan_oracle as Orc
    from tg_hip import engine as E
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    G, D = PConvUNet(), Discriminator()
    crit = InpaintingLoss(0.1, 0.1, device=torch.device("cpu"))
    G, D, crit = G.to(dev), D.to(dev), crit.to(dev)
    oG, oD = torch.optim.Adam(G.parameters(), lr=2e-4), torch.optim.Adam(D.parameters(), lr=2e-4)
    real, mask = Orc.synth_batch(args.batch, args.size, args.seed)
    out = {}

    def probe(kind, name, t):
        key = f"{kind}/{name}"
        if key in out or key + "/full" in out or key + "/sum" in out:
            return
        t = t.detach()
        td = t.double()
        out[key + "/sum"] = np.float64(td.sum().item())
        out[key + "/abssum"] = np.float64(td.abs().sum().item())
        if t.numel() <= 600000:
            out[key + "/full"] = t.float().cpu().numpy()
        else:
            flat = t.reshape(-1)
            stride = max(1, flat.numel() // 65536)
            out[key + "/sample"] = flat[::stride][:65536].float().cpu().numpy()
            out[key + "/stride"] = np.int64(stride)
            if t.dim() == 4 and t.shape[1] % 8 == 0 and t.shape[2] % 8 == 0:
                B, H, W, C = t.shape
                out[key + "/blocks"] = td.sum(3).reshape(B, H // 8, 8, W // 8, 8).sum((2, 4)).cpu().numpy()
            elif t.dim() == 3 and t.shape[1] % 8 == 0:
                B, H, W = t.shape
                out[key + "/blocks"] = td.reshape(B, H // 8, 8, W // 8, 8).sum((2, 4)).cpu().numpy()

    E.PROBE = probe
    train_step(G, D, crit, oG, oD, real.to(dev), mask.to(dev))
    torch.cuda.synchronize()
    E.PROBE = None
    for k, p_ in G.named_parameters():
        if p_.requires_grad and p_.numel() <= 4096:
            out[f"grad/{k}/full"] = p_.grad.detach().float().cpu().numpy()
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, len(out), "arrays")


def compare(a, b):
    A, Bz = dict(np.load(a)), dict(np.load(b))
    keys = sorted({k.rsplit("/", 1)[0] for k in A})
    order = ["enc1", "enc2", "enc3", "enc4", "enc5", "enc6", "enc7", "dec7", "dec6", "dec5", "dec4", "dec3", "dec2", "dec1", "final", "gen"]
    def pos(k):
        kind, name = k.split("/")[0], k.split("/")[1]
        i = order.index(name) if name in order else 99
        return (0, i) if kind == "fwd" else (1, -i) if kind == "bwd" else (2, 0)
    for k in sorted(keys, key=pos):
        line = f"{k:34s}"
        for suf in ("full", "sample", "blocks"):
            if k + "/" + suf in A:
                x, y = A[k + "/" + suf].astype(np.float64), Bz[k + "/" + suf].astype(np.float64)
                d = np.abs(x - y)
                scale = max(np.abs(x).max(), 1e-300)
                line += f"  {suf}: max|d|/max {d.max() / scale:9.2e} rms(d)/rms {np.sqrt((d ** 2).mean()) / max(np.sqrt((x ** 2).mean()), 1e-300):9.2e}"
                if suf == "blocks":
                    i = np.unravel_index(d.argmax(), d.shape)
                    line += f" worst block {i}"
        if k + "/sum" in A:
            line += f"  sum rel {abs(A[k + '/sum'] - Bz[k + '/sum']) / max(A[k + '/abssum'], 1e-300):9.2e}"
        print(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/chain_dump.npz")
    ap.add_argument("--seed", type=int, default=500)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--compare", nargs=2, default=None)
    args = ap.parse_args()
    if args.compare:
        compare(*args.compare)
    else:
        dump(args)


if __name__ == "__main__":
    main()
