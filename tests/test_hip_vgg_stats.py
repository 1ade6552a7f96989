"""The VGG trunk's Winograd F(4x4,3x3) path under TRAINED-LIKE weight statistics (tests/vgg_like.py): the round-3 verdict's
item 6.  (i) per layer, F(4x4,3x3) and F(2x2,3x3) against fp64 on the trunk's own activations; (ii) the perceptual term and its
input gradient against the fp64 oracle; (iii) the admission check InpaintingLoss runs on any non-stand-in trunk
(_admit_wino4) decides on these weights, and its decision is what the loss then uses."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tg_hip import lib
    lib.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def trained_like(tmp_path_factory):
    from tests.vgg_like import trained_like_state
    sd = trained_like_state()
    path = str(tmp_path_factory.mktemp("vgg") / "vgg16_trained_like.pth")
    torch.save({f"features.{k}": v for k, v in sd.items()}, path)
    return sd, path


def _dump(name, obj):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out", "parity")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name + ".json"), "w") as f:
        json.dump(obj, f, indent=1)


def test_trunk_layers_wino44_vs_wino22_vs_fp64(dev, trained_like):
    """Layer by layer on the trunk's OWN activations (fp64 chain on the CPU feeds every layer its exact input): error of the
    F(4x4,3x3) and F(2x2,3x3) kernels, relative to the layer's largest output and to its rms -- for the trained-like weights and
    for the default-initialised stand-in.  Recorded (gpurun_out/parity/vgg_trained_like_layers.json); asserted: F(4x4) stays
    within 3e-5 of the layer's largest output on both weight families (1e-5 on random data, tools/conv_fuzz.py)."""
    from oracle import terragan_oracle as Orc
    from tg_hip import ops as O
    from tests.vgg_like import TRUNK
    sd_t, _path = trained_like
    torch.manual_seed(0)
    Orc.init_generator(), Orc.init_discriminator()
    sd_s = Orc.init_vgg_standin()
    real, _m = Orc.synth_batch(4, 128, 77)
    report = {}
    for fam, sd in (("trained_like", sd_t), ("standin", sd_s)):
        h = real.repeat(1, 3, 1, 1).double()
        rows = {}
        for item in TRUNK:
            if item == "M":
                h = F.max_pool2d(h, 2, 2)
                continue
            w, b = sd[f"{item}.weight"], sd[f"{item}.bias"]
            ref = F.relu(F.conv2d(h, w.double(), b.double(), 1, 1))
            if item != 0 and max(w.shape[0], w.shape[1]) >= 128:
                x = h.float().permute(0, 2, 3, 1).contiguous().to(dev)
                wd = w.contiguous(memory_format=torch.channels_last).to(dev)
                refn = ref.permute(0, 2, 3, 1)
                row = {"max_out": float(refn.abs().max()), "rms_out": float(refn.pow(2).mean().sqrt()),
                       "sparsity": float((refn == 0).double().mean())}
                for nm, w4 in (("wino44", True), ("wino22", False)):
                    y = O.conv_fwd(x, wd, b.to(dev), 3, 1, 1, act=O.ACT_RELU, wino4=w4).cpu().double()
                    d = (y - refn).abs()
                    row[nm + "_max_err_over_max"] = float(d.max()) / row["max_out"]
                    row[nm + "_rms_err_over_rms"] = float(d.pow(2).mean().sqrt()) / row["rms_out"]
                rows[f"conv{item}"] = row
                assert row["wino44_max_err_over_max"] <= 3e-5, (fam, item, row)
            h = ref
        report[fam] = rows
    print(json.dumps(report, indent=1))
    _dump("vgg_trained_like_layers", report)


def test_perceptual_gradient_default_path_at_cpu_fp32_level(dev, trained_like):
    """The perceptual term (losses.py:79-90) and d/dpred on a batch that differs from its target inside the holes only -- the
    train step's situation -- against the fp64 oracle, for the trained-like trunk and the stand-in, at two distances between
    prediction and target.  The gradient is sign(fp - ft) pushed back through the trunk, so forward rounding error flips signs
    wherever |fp - ft| is small: the DEFAULT path (forward F(2x2,3x3), dgrad F(4x4,3x3)) must sit at the CPU fp32 evaluation's
    own error (<= 2x its rms error + 0.5 %), loss within rtol 1e-6 + 5 x the CPU deviation.  F(4x4,3x3) in the FORWARD is
    recorded next to it: 21-33 % rms gradient error -- why it is off (tg_hip/engine.py)."""
    from oracle import terragan_oracle as Orc
    from tg_hip import engine as E
    from tg_hip import ops as O
    sd_t, _path = trained_like
    torch.manual_seed(0)
    Orc.init_generator(), Orc.init_discriminator()
    fams = {"trained_like": sd_t, "standin": Orc.init_vgg_standin()}
    B, size = 8, 256                                         # 8 x 2 images: the F(4x4) work items fill the chip (>= 256 items)
    real, mask = Orc.synth_batch(B, size, 91)
    noise = torch.rand(real.shape, generator=torch.Generator().manual_seed(92))
    rep = {}
    for fam, sd in fams.items():
        V = {k: v.to(dev) for k, v in sd.items()}
        for k in list(V):
            if k.endswith(".weight"):
                V[k] = O.weight_view(V[k].contiguous(memory_format=torch.channels_last)).permute(0, 3, 1, 2)
        V["0.folded"] = O.fold_cin(V["0.weight"])
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
            cpu_l = abs(res[torch.float32][0] - l64) / abs(l64)
            cpu_g = float((res[torch.float32][1] - g64).pow(2).mean().sqrt()) / grms
            row = {"cpu_fp32": {"loss_rel": cpu_l, "grad_rms": cpu_g}}
            both = torch.cat([pred, real]).reshape(2 * B, size, size).to(dev).contiguous()
            for nm, fmode, bmode in (("default", None, None), ("fwd_wino22_bwd_wino22", False, False), ("fwd_wino44_bwd_wino44", True, True)):
                feats, ctx = E.vgg_forward(V, both, keep=True, wino4=fmode)
                perc, dfeat = O.l1_mean(feats[:B], feats[B:], 1.0, want_grad=True)
                dp = E.vgg_backward(ctx, dfeat, nb=B, wino4=bmode).cpu().double().reshape(g64.shape)
                row[nm] = {"loss_rel": abs(float(perc) - l64) / abs(l64), "grad_rms": float((dp - g64).pow(2).mean().sqrt()) / grms}
            rep[f"{fam}/blend{blend}"] = row
            d = row["default"]
            assert d["loss_rel"] <= 1e-6 + 5 * cpu_l, (fam, blend, row)
            assert d["grad_rms"] <= 2 * cpu_g + 5e-3, (fam, blend, row)
    print(json.dumps(rep, indent=1))
    _dump("vgg_perceptual_gradient", rep)
