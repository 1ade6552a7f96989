#!/usr/bin/env python3
"""(Round 4: the measurement that took Winograd F(4x4,3x3) out of the VGG trunk's FORWARD -- tg_hip/engine.py.)
How the perceptual term's gradient degrades as the prediction approaches the target, per trunk weight family and per
Winograd variant: d/dpred mean|VGG(pred) - VGG(target)| is sign(fp - ft) pushed back through the trunk, so a kernel's rounding
error matters in proportion to |fp - ft| -- which shrinks as training converges.  Prints, for pred = target outside the holes
and target + blend * (noise - target) inside, the error of F(4x4,3x3) and F(2x2,3x3) against the fp64 oracle (CPU)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "terra-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("TERRAGAN_ALLOW_STANDIN_VGG", "1")
import torch  # noqa: E402


def main():
    from oracle import terragan_oracle as Orc
    from tests.vgg_like import trained_like_state
    from tg_hip import engine as E
    from tg_hip import ops as O
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    Orc.init_generator(), Orc.init_discriminator()
    fams = {"standin": Orc.init_vgg_standin(), "trained_like": trained_like_state()}
    B, size = 8, 256
    real, mask = Orc.synth_batch(B, size, 91)
    noise = torch.rand(real.shape, generator=torch.Generator().manual_seed(92))
    out = {}
    for fam, sd in fams.items():
        V = {k: v.to(dev) for k, v in sd.items()}
        for k in list(V):
            if k.endswith(".weight"):
                V[k] = O.weight_view(V[k].contiguous(memory_format=torch.channels_last)).permute(0, 3, 1, 2)
        V["0.folded"] = O.fold_cin(V["0.weight"])
        for blend in (0.4, 0.1, 0.03, 0.01):
            pred = (real * mask + (real + blend * (noise - real)) * (1 - mask)).contiguous()
            res = {}
            for dt in (torch.float64, torch.float32):
                p = pred.to(dt).requires_grad_(True)
                q = {k: v.to(dt) for k, v in sd.items()}
                loss = (Orc.vgg_features(q, p) - Orc.vgg_features(q, real.to(dt))).abs().mean()
                (gr,) = torch.autograd.grad(loss, p)
                res[dt] = (float(loss.detach()), gr.double())
            l64, g64 = res[torch.float64]
            grms = float(g64.pow(2).mean().sqrt())
            row = {"cpu_fp32": {"loss_rel": abs(res[torch.float32][0] - l64) / abs(l64),
                                "grad_rms": float((res[torch.float32][1] - g64).pow(2).mean().sqrt()) / grms}}
            both = torch.cat([pred, real]).reshape(2 * B, size, size).to(dev).contiguous()
            for nm, mode, bmode in (("wino44", True, True), ("wino22", False, False), ("fwd22_bwd44", False, True), ("fwd44_bwd22", True, False)):
                feats, ctx = E.vgg_forward(V, both, keep=True, wino4=mode)
                perc, dfeat = O.l1_mean(feats[:B], feats[B:], 1.0, want_grad=True)
                dp = E.vgg_backward(ctx, dfeat, nb=B, wino4=bmode).cpu().double().reshape(g64.shape)
                row[nm] = {"loss_rel": abs(float(perc) - l64) / abs(l64), "grad_rms": float((dp - g64).pow(2).mean().sqrt()) / grms}
            out[f"{fam}/blend{blend}"] = row
            print(fam, blend, json.dumps(row), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
