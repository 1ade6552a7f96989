#!/usr/bin/env python3
"""PROTOTYPE (numerics only, not the product path): the perceptual term by DIFFERENCE PROPAGATION.

mean|VGG(pred) - VGG(target)| needs sign(fp - ft), and pred = target outside the holes: fp - ft is tiny against the features, so a
forward kernel's rounding error (relative to |f|) decides the sign -- which is what took Winograd F(4x4,3x3) out of the trunk's forward
(DESIGN 2b).  The convolutions are linear: z_p - z_t = conv(a_p - a_t).  Carrying (a_t, delta = a_p - a_t) through the trunk instead of
(a_p, a_t) makes every rounding error RELATIVE TO THE DIFFERENCE:
    conv:     z_t = conv(a_t) + b,   dz = conv(delta)                       (any kernel; its error scales with |delta|)
    ReLU:     delta' = z_t > 0 ? (z_t + dz > 0 ? dz : -z_t) : (z_t + dz > 0 ? z_t + dz : 0)
    max-pool: delta' = a_t[i_p] + delta[i_p] - a_t[i_t],   i_p / i_t = arg max of a_t + delta / of a_t in the window
    loss:     mean |delta_last|,   backward through the pred branch with the gates of a_p = a_t + delta.
This script runs that chain with the library's conv kernels (F(4x4,3x3) forced in the forward) and torch element-wise ops in between,
and prints the error of d loss / d pred against the fp64 oracle next to the plain F(4x4) / F(2x2) forward and the CPU fp32 oracle."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "terra-gan_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("TERRAGAN_ALLOW_STANDIN_VGG", "1")
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


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
    for fam, sd in fams.items():
        V = {k: v.to(dev) for k, v in sd.items()}
        for k in list(V):
            if k.endswith(".weight"):
                V[k] = O.weight_view(V[k].contiguous(memory_format=torch.channels_last)).permute(0, 3, 1, 2)
        V["0.folded"] = O.fold_cin(V["0.weight"])
        zero_b = {k: torch.zeros_like(v) for k, v in V.items() if k.endswith(".bias")}
        for blend in (0.4, 0.03):
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
            row = {"cpu_fp32": float((res[torch.float32][1] - g64).pow(2).mean().sqrt()) / grms}
            both = torch.cat([pred, real]).reshape(2 * B, size, size).to(dev).contiguous()
            for nm, mode in (("plain_F44", True), ("plain_F22", False)):
                feats, ctx = E.vgg_forward(V, both, keep=True, wino4=mode)
                perc, dfeat = O.l1_mean(feats[:B], feats[B:], 1.0, want_grad=True)
                dp = E.vgg_backward(ctx, dfeat, nb=B, wino4=True).cpu().double().reshape(g64.shape)
                row[nm] = float((dp - g64).pow(2).mean().sqrt()) / grms
            # ---- difference propagation, forward convs on F(4x4,3x3) wherever the geometry allows ----
            for nm, mode in (("delta_F44", True), ("delta_F22", False)):
                t = real.reshape(B, size, size, 1).to(dev).contiguous()
                d = (pred - real).reshape(B, size, size, 1).to(dev).contiguous()          # exact where pred == target
                steps = []
                for item in E.VGG_TRUNK:
                    if item == "M":
                        ap = t + d
                        Bn, H, W, C = t.shape
                        tw = t.reshape(Bn, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(Bn, H // 2, W // 2, C, 4)
                        dw = d.reshape(Bn, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4).reshape(Bn, H // 2, W // 2, C, 4)
                        it_ = tw.argmax(-1, keepdim=True)
                        ip_ = (tw + dw).argmax(-1, keepdim=True)
                        t_new = tw.gather(-1, it_).squeeze(-1)
                        d_new = (tw.gather(-1, ip_).squeeze(-1) - t_new) + dw.gather(-1, ip_).squeeze(-1)
                        steps.append(("M", ap))
                        t, d = t_new.contiguous(), d_new.contiguous()
                        continue
                    w = V["0.folded"] if item == 0 else V[f"{item}.weight"]
                    w4 = E._vgg_wino4(w, t.shape[0], t.shape[1], t.shape[2], w.shape[0], mode, False)
                    zt = O.conv_fwd(t, w, V[f"{item}.bias"], 3, 1, 1, act=O.ACT_NONE, wino4=w4)
                    dz = O.conv_fwd(d, w, zero_b[f"{item}.bias"], 3, 1, 1, act=O.ACT_NONE, wino4=w4)
                    zp = zt + dz
                    dnew = torch.where(zt > 0, torch.where(zp > 0, dz, -zt), torch.where(zp > 0, zp, torch.zeros_like(zp)))
                    t = zt.clamp_min(0)
                    steps.append(("C", w, tuple(d.shape), (t + dnew)))                    # a_p = a_t + delta: the backward's gate
                    d = dnew.contiguous()
                n = d.numel()
                dfeat = torch.sign(d) / n
                # backward through the pred branch exactly as engine.vgg_backward does (gates of a_p)
                ctx = E.NS(steps=[E.NS(kind="M", x=s[1]) if s[0] == "M" else E.NS(kind="C", w=s[1], x_shape=s[2], a=s[3]) for s in steps])
                dp = E.vgg_backward(ctx, dfeat.contiguous(), nb=None, wino4=True).cpu().double().reshape(g64.shape)
                row[nm] = float((dp - g64).pow(2).mean().sqrt()) / grms
                row[nm + "_loss_rel"] = abs(float(d.abs().mean()) - l64) / abs(l64)
            print(fam, blend, json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
