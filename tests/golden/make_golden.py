#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; the GPU box has none).
The reference's own modules (mvp_gan/src/models/{pconv,generator,discriminator}.py and
mvp_gan/src/utils/losses.py) are imported by file path under a synthetic parent package
(SURVEY.md §8c recipe); `torchvision` is absent here, so a stub exposes `vgg16(weights=...)`
returning the standard VGG16-D `features` Sequential with nn.Conv2d default init (deterministic
stand-in weights: the ImageNet weights are not fetchable offline).

Nothing of the reference's source is written out: fixtures hold inputs (or the seed that makes
them) and expected outputs only.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import importlib.util
import io
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.terragan_oracle import synth_batch  # noqa: E402  (input recipe only)

logging.disable(logging.CRITICAL)
torch.set_num_threads(8)


def _load_reference():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")

    class VGG16_Weights:  # noqa: N801
        IMAGENET1K_V1 = "standin"

    def vgg16(weights=None):
        cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
        layers, cin = [], 3
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        m = nn.Module()
        m.features = nn.Sequential(*layers)
        return m

    tvm.vgg16, tvm.VGG16_Weights = vgg16, VGG16_Weights
    tv.models = tvm
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tvm

    for pkg in ["refpkg", "refpkg.models", "refpkg.utils"]:
        m = types.ModuleType(pkg)
        m.__path__ = []
        sys.modules[pkg] = m

    def load(modname, rel):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    pconv = load("refpkg.models.pconv", "mvp_gan/src/models/pconv.py")
    gen = load("refpkg.models.generator", "mvp_gan/src/models/generator.py")
    disc = load("refpkg.models.discriminator", "mvp_gan/src/models/discriminator.py")
    losses = load("refpkg.utils.losses", "mvp_gan/src/utils/losses.py")
    return pconv, gen, disc, losses


def summ(t, full_limit=20000):
    """Full tensor when small, else (sum, abs-sum, l2, strided sample) in float64."""
    t = t.detach().double().flatten()
    if t.numel() <= full_limit:
        return {"full": t.float().numpy()}
    stride = max(1, t.numel() // 512)
    return {"sum": np.float64(t.sum()), "abssum": np.float64(t.abs().sum()),
            "l2": np.float64(t.norm()), "stride": np.int64(stride),
            "sample": t[::stride][:512].float().numpy()}


def put(out, name, t, **kw):
    for k, v in summ(t, **kw).items():
        out[f"{name}/{k}"] = v


def mask_case(kind, b, h, w, g):
    m = torch.ones(b, 1, h, w)
    if kind == "ones":
        pass
    elif kind == "zeros":
        m.zero_()
    elif kind == "holes30":
        m = (torch.rand(b, 1, h, w, generator=g) > 0.3).float()
    elif kind == "single":
        m.zero_()
        m[:, :, h // 2, w // 3] = 1
    elif kind == "border":
        m.zero_()
        m[:, :, 0, :] = 1
        m[:, :, -1, :] = 1
        m[:, :, :, 0] = 1
        m[:, :, :, -1] = 1
    elif kind == "blocks":
        m = (torch.rand(b, 1, (h + 3) // 4, (w + 3) // 4, generator=g) > 0.4).float()
        m = m.repeat_interleave(4, 2).repeat_interleave(4, 3)[:, :, :h, :w].contiguous()
    return m


PCONV_CASES = [  # (cin, cout, k, s, p, B, H, W)
    (1, 64, 7, 2, 3, 2, 24, 24),
    (64, 128, 5, 2, 2, 2, 24, 24),
    (256, 512, 3, 2, 1, 2, 12, 12),
    (192, 64, 3, 1, 1, 2, 16, 16),
    (64, 64, 3, 1, 1, 2, 24, 20),
    (32, 32, 3, 2, 1, 3, 9, 11),      # odd sizes
]
MASK_KINDS = ["ones", "zeros", "holes30", "single", "border", "blocks"]


def gen_pconv(ref_pconv):
    out = {}
    for ci, (cin, cout, k, s, p, b, h, w) in enumerate(PCONV_CASES):
        for kind in MASK_KINDS:
            tag = f"c{ci}_{kind}"
            torch.manual_seed(100 + ci)
            layer = ref_pconv.PConv2d(cin, cout, k, s, p)
            with torch.no_grad():          # non-trivial BN affine
                layer.bn.weight.uniform_(0.5, 1.5)
                layer.bn.bias.uniform_(-0.3, 0.3)
            g = torch.Generator().manual_seed(200 + ci)
            x = torch.randn(b, cin, h, w, generator=g, requires_grad=True)
            m = mask_case(kind, b, h, w, g)
            y, mo = layer(x, m)
            gy = torch.randn(y.shape, generator=g)
            (y * gy).sum().backward()
            out[f"{tag}/cfg"] = np.array([cin, cout, k, s, p, b, h, w], dtype=np.int64)
            put(out, f"{tag}/y", y)
            put(out, f"{tag}/mask_out", mo)
            put(out, f"{tag}/dx", x.grad)
            put(out, f"{tag}/dw", layer.input_conv.weight.grad)
            put(out, f"{tag}/db", layer.input_conv.bias.grad)
            put(out, f"{tag}/dgamma", layer.bn.weight.grad)
            put(out, f"{tag}/dbeta", layer.bn.bias.grad)
            put(out, f"{tag}/running_mean", layer.bn.running_mean)
            put(out, f"{tag}/running_var", layer.bn.running_var)
            layer.eval()
            with torch.no_grad():
                ye, _ = layer(x, m)
            put(out, f"{tag}/y_eval", ye)
    return out


# g64 (B = 2): BatchNorm over TWO values at the 1x1 bottleneck -- x_hat = +-1, the gradient through it is rounding noise times
# a huge rstd: a plumbing case (shapes, odd sizes).  g64b16: the same geometry with 16 values per channel at the bottleneck.
# g72x40b3: the odd-size case (crop / pad in _pad_to_match, generator.py:78-84, forward AND backward) with THREE values per channel
# at the bottleneck, so that its gradients can be held to values (the B = 2 cases cannot: see test_generator_golden)
MODEL_G_CASES = [("g64", 2, 64, 64), ("g72x40", 2, 72, 40), ("g96", 3, 96, 96), ("g64b16", 16, 64, 64), ("g72x40b3", 3, 72, 40)]


def gen_models(ref_gen, ref_disc):
    """Whole-generator / discriminator forward+backward at small sizes, incl. an odd size that
    exercises _pad_to_match (generator.py:78-84)."""
    out = {}
    for tag, b, h, w in MODEL_G_CASES:
        torch.manual_seed(7)
        G = ref_gen.PConvUNet()
        x, m = synth_batch(b, max(h, w), 300 + h)
        x, m = x[:, :, :h, :w].contiguous(), m[:, :, :h, :w].contiguous()
        xm = (x * m).requires_grad_(True)
        y = G(xm, m)
        g = torch.Generator().manual_seed(5)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        out[f"{tag}/cfg"] = np.array([b, h, w], dtype=np.int64)
        put(out, f"{tag}/out", y)
        put(out, f"{tag}/dx", xm.grad)
        for n, p_ in G.named_parameters():
            if p_.grad is not None:
                put(out, f"{tag}/grad/{n}", p_.grad, full_limit=2048)
        for n, buf in G.named_buffers():
            if "running" in n:
                put(out, f"{tag}/buf/{n}", buf, full_limit=2048)
        G.eval()
        with torch.no_grad():
            put(out, f"{tag}/out_eval", G(xm.detach(), m))
    for tag, b, h, w in [("d64", 2, 64, 64), ("d80x48", 3, 80, 48)]:
        torch.manual_seed(8)
        D = ref_disc.Discriminator()
        g = torch.Generator().manual_seed(6)
        x = torch.rand(b, 1, h, w, generator=g, requires_grad=True)
        y = D(x)
        gy = torch.randn(y.shape, generator=g)
        (y * gy).sum().backward()
        out[f"{tag}/cfg"] = np.array([b, h, w], dtype=np.int64)
        put(out, f"{tag}/out", y)
        put(out, f"{tag}/dx", x.grad)
        for n, p_ in D.named_parameters():
            put(out, f"{tag}/grad/{n}", p_.grad, full_limit=2048)
        for n, buf in D.named_buffers():
            if "running" in n:
                put(out, f"{tag}/buf/{n}", buf, full_limit=2048)
    return out


def gen_losses(ref_losses):
    out = {}
    torch.manual_seed(11)
    crit = ref_losses.InpaintingLoss(0.1, 0.1, device=torch.device("cpu"))
    g = torch.Generator().manual_seed(12)
    for tag, b, h, w, kind in [("l32", 2, 32, 32, "blocks"), ("l32ones", 2, 32, 32, "ones"),
                               ("l48x40", 3, 48, 40, "holes30"), ("l32zeros", 1, 32, 32, "zeros")]:
        pred = torch.rand(b, 1, h, w, generator=g, requires_grad=True)
        tgt = torch.rand(b, 1, h, w, generator=g)
        m = mask_case(kind, b, h, w, g)
        out[f"{tag}/cfg"] = np.array([b, h, w], dtype=np.int64)
        out[f"{tag}/pred"] = pred.detach().numpy()
        out[f"{tag}/target"] = tgt.numpy()
        out[f"{tag}/mask"] = m.numpy().astype(np.uint8)
        total = crit(pred, tgt, m)
        total.backward()
        put(out, f"{tag}/total", total)
        put(out, f"{tag}/dpred", pred.grad)
        with torch.no_grad():
            put(out, f"{tag}/l1", crit.l1_loss(pred, tgt))
            put(out, f"{tag}/tv", crit.total_variation_loss(pred * (1 - m)))
            put(out, f"{tag}/boundary", crit.boundary_loss(pred, tgt, m))
            put(out, f"{tag}/perc", crit.l1_loss(crit.vgg_layers(pred.repeat(1, 3, 1, 1)),
                                                 crit.vgg_layers(tgt.repeat(1, 3, 1, 1))))
    # human-guided loss (losses.py:132-204)
    cfg = {"training": {"loss_weights": {"boundary": 0.5},
                        "modes": {"human_guided": {"human_feedback_weight": 0.3, "base_loss_weight": 0.7}}}}
    torch.manual_seed(11)
    hcrit = ref_losses.HumanGuidedLoss(cfg, device=torch.device("cpu"))
    pred = torch.rand(2, 1, 32, 32, generator=g, requires_grad=True)
    tgt = torch.rand(2, 1, 32, 32, generator=g)
    m = mask_case("blocks", 2, 32, 32, g)
    hm = mask_case("blocks", 2, 32, 32, g) * 255.0
    out["hg/pred"], out["hg/target"] = pred.detach().numpy(), tgt.numpy()
    out["hg/mask"], out["hg/human"] = m.numpy().astype(np.uint8), hm.numpy().astype(np.uint8)
    tot = hcrit(pred, tgt, m, {"mask": hm})
    tot.backward()
    put(out, "hg/total", tot)
    put(out, "hg/dpred", pred.grad)
    # BCE-with-logits, constant targets (train.py:115,203,215-216)
    bce = nn.BCEWithLogitsLoss()
    z = (torch.randn(2, 1, 15, 15, generator=g) * 3).requires_grad_(True)
    out["bce/logits"] = z.detach().numpy()
    for tv_, nm in [(1.0, "one"), (0.0, "zero")]:
        z.grad = None
        l_ = bce(z, torch.full_like(z, tv_))
        l_.backward()
        put(out, f"bce/{nm}", l_)
        put(out, f"bce/d{nm}", z.grad)
    return out


def _ref_step(G, D, crit, bce, oG, oD, real, mask):
    """The reference loop body, train.py:177-219, driven verbatim in order."""
    masked = real * mask
    oG.zero_grad()
    gen = G(masked, mask)
    g_loss = crit(gen, real, mask)
    fv = D(gen)
    g_adv = bce(fv, torch.ones_like(fv))
    g_total = g_loss + g_adv
    g_total.backward()
    ggrads = {n: p.grad.clone() for n, p in G.named_parameters() if p.grad is not None}
    oG.step()
    oD.zero_grad()
    rv = D(real)
    fv = D(gen.detach())
    real_loss = bce(rv, torch.ones_like(rv))
    fake_loss = bce(fv, torch.zeros_like(fv))
    d_loss = 0.5 * (real_loss + fake_loss)
    d_loss.backward()
    dgrads = {n: p.grad.clone() for n, p in D.named_parameters() if p.grad is not None}
    oD.step()
    sc = dict(g_total=g_total, g_loss=g_loss, g_adv=g_adv, d_loss=d_loss, real_loss=real_loss,
              fake_loss=fake_loss)
    return gen.detach(), {k: float(v) for k, v in sc.items()}, ggrads, dgrads


def _build(ref_gen, ref_disc, ref_losses, seed=0):
    torch.manual_seed(seed)
    G, D = ref_gen.PConvUNet(), ref_disc.Discriminator()
    crit = ref_losses.InpaintingLoss(0.1, 0.1, device=torch.device("cpu"))
    oG = torch.optim.Adam(G.parameters(), lr=2e-4)
    oD = torch.optim.Adam(D.parameters(), lr=2e-4)
    return G, D, crit, oG, oD


def gen_steps(ref_gen, ref_disc, ref_losses):
    out = {}
    bce = nn.BCEWithLogitsLoss()
    for tag, b, size, nsteps, seed0 in [("c1_256", 1, 256, 3, 1), ("b4_128", 4, 128, 3, 40)]:
        G, D, crit, oG, oD = _build(ref_gen, ref_disc, ref_losses)
        G.train(), D.train()
        out[f"{tag}/cfg"] = np.array([b, size, nsteps, seed0], dtype=np.int64)
        for s in range(nsteps):
            real, mask = synth_batch(b, size, seed0 + s)
            gen, sc, gg, dg = _ref_step(G, D, crit, bce, oG, oD, real, mask)
            for k, v in sc.items():
                out[f"{tag}/s{s}/{k}"] = np.float64(v)
            put(out, f"{tag}/s{s}/gen", gen, full_limit=70000)
            if s == 0:
                for n, t in gg.items():
                    put(out, f"{tag}/s0/ggrad/{n}", t, full_limit=64)
                for n, t in dg.items():
                    put(out, f"{tag}/s0/dgrad/{n}", t, full_limit=64)
            if s in (0, nsteps - 1):
                for n, p_ in list(G.named_parameters()) + list(D.named_parameters()):
                    pre = "G" if any(p_ is q for q in G.parameters()) else "D"
                    out[f"{tag}/s{s}/w/{pre}.{n}"] = np.array(
                        [float(p_.double().sum()), float(p_.double().abs().sum())])
                for n, buf in list(G.named_buffers()) + list(D.named_buffers()):
                    if "running" in n and n.split(".")[0] in ("enc1", "enc7", "dec1", "model"):
                        put(out, f"{tag}/s{s}/buf/{n}", buf, full_limit=1024)
    out.update(gen_dp(ref_gen, ref_disc, ref_losses, [("dp2_128", 2, 4, 128)]))
    return out


FULL_CASES = [("c2_b16_256", 16, 256, 1, 500), ("c3_b8_512", 8, 512, 1, 600)]      # BASELINE configs[1] / configs[2] shapes


def gen_steps_full(ref_gen, ref_disc, ref_losses):
    """One reference train step (train.py:177-219) at the HEADLINE sizes: B=16 at 256^2 (BASELINE configs[1]) and B=8 at
    512^2 (configs[2]'s shape, fp32).  Losses, a strided sample + sums of the generated batch, per-tensor gradient
    summaries, post-Adam weight sums and the BatchNorm buffers of a few layers."""
    out = {}
    bce = nn.BCEWithLogitsLoss()
    for tag, b, size, nsteps, seed0 in FULL_CASES:
        G, D, crit, oG, oD = _build(ref_gen, ref_disc, ref_losses)
        G.train(), D.train()
        out[f"{tag}/cfg"] = np.array([b, size, nsteps, seed0], dtype=np.int64)
        real, mask = synth_batch(b, size, seed0)
        gen, sc, gg, dg = _ref_step(G, D, crit, bce, oG, oD, real, mask)
        for k, v in sc.items():
            out[f"{tag}/s0/{k}"] = np.float64(v)
        put(out, f"{tag}/s0/gen", gen, full_limit=4096)
        for n, t in gg.items():
            put(out, f"{tag}/s0/ggrad/{n}", t, full_limit=64)
        for n, t in dg.items():
            put(out, f"{tag}/s0/dgrad/{n}", t, full_limit=64)
        for n, p_ in list(G.named_parameters()) + list(D.named_parameters()):
            pre = "G" if any(p_ is q for q in G.parameters()) else "D"
            out[f"{tag}/s0/w/{pre}.{n}"] = np.array([float(p_.double().sum()), float(p_.double().abs().sum())])
        for n, buf in list(G.named_buffers()) + list(D.named_buffers()):
            if "running" in n and n.split(".")[0] in ("enc1", "enc7", "dec1", "model"):
                put(out, f"{tag}/s0/buf/{n}", buf, full_limit=1024)
        print(f"  {tag}: done", flush=True)
    return out


def gen_dp(ref_gen, ref_disc, ref_losses, cases):
    """Data-parallel emulation: N micro-batches, identical weights, mean grads, one Adam."""
    import copy
    out = {}
    bce = nn.BCEWithLogitsLoss()
    for tag, n, b, size in cases:
        G, D, crit, oG, oD = _build(ref_gen, ref_disc, ref_losses)
        out[f"{tag}/cfg"] = np.array([n, b, size], dtype=np.int64)
        gens, gsum = [], {}
        reps = [(G, D)] + [(copy.deepcopy(G), copy.deepcopy(D)) for _ in range(n - 1)]
        batches = [synth_batch(b, size, 1000 + r) for r in range(n)]
        for r, (Gr, Dr) in enumerate(reps):
            Gr.train(), Dr.train()
            real, mask = batches[r]
            Gr.zero_grad()
            gen = Gr(real * mask, mask)
            fv = Dr(gen)
            tot = crit(gen, real, mask) + bce(fv, torch.ones_like(fv))
            tot.backward()
            out[f"{tag}/r{r}/g_total"] = np.float64(float(tot))
            gens.append(gen.detach())
            for (nm, p_) in Gr.named_parameters():
                if p_.grad is not None:
                    gsum[nm] = gsum.get(nm, 0) + p_.grad / n
        for nm, p_ in G.named_parameters():
            if nm in gsum:
                p_.grad = gsum[nm].clone()
        oG.step()
        dsum = {}
        for r, (Gr, Dr) in enumerate(reps):
            real, mask = batches[r]
            Dr.zero_grad()
            rv, fv = Dr(real), Dr(gens[r])
            dl = 0.5 * (bce(rv, torch.ones_like(rv)) + bce(fv, torch.zeros_like(fv)))
            dl.backward()
            out[f"{tag}/r{r}/d_loss"] = np.float64(float(dl))
            for (nm, p_) in Dr.named_parameters():
                dsum[nm] = dsum.get(nm, 0) + p_.grad / n
        for nm, p_ in D.named_parameters():
            p_.grad = dsum[nm].clone()
        oD.step()
        for nm, t in gsum.items():
            put(out, f"{tag}/ggrad/{nm}", t, full_limit=64)
        for nm, t in dsum.items():
            put(out, f"{tag}/dgrad/{nm}", t, full_limit=64)
        for nm, p_ in G.named_parameters():
            out[f"{tag}/w/G.{nm}"] = np.array([float(p_.double().sum()), float(p_.double().abs().sum())])
        for nm, p_ in D.named_parameters():
            out[f"{tag}/w/D.{nm}"] = np.array([float(p_.double().sum()), float(p_.double().abs().sum())])
    return out


def gen_init(ref_gen, ref_disc, ref_losses):
    """Seeded-init parity + state-dict key contract (SURVEY §5 'Checkpoint', §8b)."""
    out = {}
    G, D, crit, _, _ = _build(ref_gen, ref_disc, ref_losses)
    for pre, mod in [("G", G), ("D", D), ("V", crit.vgg_layers)]:
        sd = mod.state_dict()
        out[f"{pre}/keys"] = np.array(list(sd.keys()))
        out[f"{pre}/shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
        for k, v in sd.items():
            if v.dtype.is_floating_point:
                out[f"{pre}/w/{k}"] = np.array([float(v.double().sum()), float(v.double().abs().sum())])
                out[f"{pre}/first/{k}"] = v.flatten()[:8].numpy()
    return out


def gen_steps_unc(cases=(("c1_256", 1, 256, 3, 1), ("b4_128", 4, 128, 3, 40)), with_dp=True, alts=False):
    """fp32-vs-fp64 deviation of the step fixtures' quantities, from the oracle (the reference itself cannot run in
    fp64: pconv.py:35,40 hard-code .float(); the oracle is pinned to the reference in fp32 by steps.npz).  The GPU parity
    tests bound their error by the stated fp32 tolerance + k x this deviation: a result inside the
    reference arithmetic's own fp32 uncertainty cannot be told from the reference."""
    from oracle import terragan_oracle as Orc
    out = {}

    def state(dtype):
        st = Orc.TrainState(0)
        for d in (st.gp, st.dp, st.vp):
            for k in d:
                if d[k].dtype.is_floating_point:
                    d[k] = d[k].to(dtype)
        st.opt_g = Orc.Adam(st.gp, Orc.trainable(st.gp), 2e-4)
        st.opt_d = Orc.Adam(st.dp, Orc.trainable(st.dp), 2e-4)
        return st

    def dev4(a, c):
        a, c = a.double().flatten(), c.double().flatten()
        d = a - c
        return np.array([float(d.abs().max()), float(d.abs().sum()), float(d.norm()), float(a.abs().max())])

    def run(dtype, b, size, nsteps, seed0):
        st = state(dtype)
        res = []
        for s in range(nsteps):
            real, mask = synth_batch(b, size, seed0 + s)
            gen, sc, gg, dg = Orc.train_step(st, real.to(dtype), mask.to(dtype))
            w = {f"G.{k}": st.gp[k].double().clone() for k in Orc.trainable(st.gp)}
            w.update({f"D.{k}": st.dp[k].double().clone() for k in Orc.trainable(st.dp)})
            bufs = {k: v.double().clone() for d in (st.gp, st.dp) for k, v in d.items() if "running" in k}
            res.append((gen.double(), {k: float(v) for k, v in sc.items()}, gg, dg, w, bufs))
        return res

    def put_max(key, val):
        """Keep the LARGEST deviation over the fp32 evaluations (element-wise for the 4-vectors)."""
        val = np.asarray(val, dtype=np.float64)
        out[key] = np.maximum(out[key], val) if key in out else val

    for tag, b, size, nsteps, seed0 in cases:
        r64 = run(torch.float64, b, size, nsteps, seed0)
        # Several fp32 evaluations of the SAME arithmetic that differ only in their reduction order -- oneDNN convolutions
        # (what the reference runs), ATen's native convolutions, and a single thread: a ReLU gate whose pre-activation sits
        # within rounding of zero flips between them, which moves a per-channel BatchNorm gradient of a few-row layer by
        # ~1/rows.  One fp32/fp64 pair is a single draw of that; the fixture keeps the largest deviation over the draws.
        variants = [("onednn", True, None)]
        if alts:
            variants += [("native", False, None), ("onethread", True, 1)]
        for _name, mkldnn_on, nthreads in variants:
            old_threads = torch.get_num_threads()
            if nthreads:
                torch.set_num_threads(nthreads)
            with torch.backends.mkldnn.flags(enabled=mkldnn_on):
                r32 = run(torch.float32, b, size, nsteps, seed0)
            torch.set_num_threads(old_threads)
            for s in range(nsteps):
                for k in r32[s][1]:
                    put_max(f"{tag}/s{s}/{k}", abs(r32[s][1][k] - r64[s][1][k]))
                put_max(f"{tag}/s{s}/gen", float((r32[s][0] - r64[s][0]).abs().max()))
                put_max(f"{tag}/s{s}/gen_mean", float((r32[s][0] - r64[s][0]).abs().mean()))
                if s in (0, nsteps - 1):
                    for k in r32[s][4]:
                        a, c = r32[s][4][k], r64[s][4][k]
                        put_max(f"{tag}/s{s}/w/{k}", [abs(float(a.sum() - c.sum())), abs(float(a.abs().sum() - c.abs().sum()))])
                    for k in r32[s][5]:
                        put_max(f"{tag}/s{s}/buf/{k}", float((r32[s][5][k] - r64[s][5][k]).abs().max()))
            for kind, idx in (("ggrad", 2), ("dgrad", 3)):
                for k in r32[0][idx]:
                    put_max(f"{tag}/s0/{kind}/{k}", dev4(r32[0][idx][k], r64[0][idx][k]))     # [max|d|, sum|d|, ||d||, max|g|]
            print(f"  {tag}: fp32 variant {_name} done", flush=True)
    # data-parallel emulations: N micro-batches, mean gradients, one Adam step
    for tag, n, b, size in ([("dp2_128", 2, 4, 128), ("dp8_128", 8, 4, 128)] if with_dp else []):
        batches = [synth_batch(b, size, 1000 + r) for r in range(n)]
        rr = {}
        for dtype in (torch.float32, torch.float64):
            st = state(dtype)
            _gens, scal = Orc.dp_train_step(st, [x.to(dtype) for x, _ in batches], [m.to(dtype) for _, m in batches])
            rr[dtype] = (scal, st.last_gg, st.last_dg, st)
        (s32, gg32, dg32, st32), (s64, gg64, dg64, st64) = rr[torch.float32], rr[torch.float64]
        for r in range(n):
            for k in ("g_total", "d_loss"):
                out[f"{tag}/r{r}/{k}"] = np.float64(abs(float(s32[r][k]) - float(s64[r][k])))
        for k in gg32:
            out[f"{tag}/ggrad/{k}"] = dev4(gg32[k], gg64[k])
        for k in dg32:
            out[f"{tag}/dgrad/{k}"] = dev4(dg32[k], dg64[k])
        for pre, d32, d64 in (("G", st32.gp, st64.gp), ("D", st32.dp, st64.dp)):
            for k in Orc.trainable(d32):
                a, c = d32[k].double(), d64[k].double()
                out[f"{tag}/w/{pre}.{k}"] = np.array([abs(float(a.sum() - c.sum())), abs(float(a.abs().sum() - c.abs().sum()))])
    return out


# (tag, batch, size, data seed[, mode]): mode "step" = the train step's generator loss (train.py:188-204) drives the backward;
# "gy" = a seeded smooth upstream gradient instead (sum(gen * gy), gy ~ N(0,1)/numel): the SAME forward and backward chain of
# the generator without the loss stack's discontinuities (|p - t| and LeakyReLU / ReLU / max-pool gates of D and the VGG
# trunk: where p ~ t to rounding a single boundary-loss pixel flips a gradient of 0.5 / sum(band) -- any two fp32 evaluations
# may disagree on it, the reference's own CPU evaluations included)
CHAIN_CASES = [("c2_b16_256", 16, 256, 500)]
CHAIN_GY_CASES = [(f"c2_b16_256_gy_s{s}", 16, 256, s, "gy") for s in (500, 501, 502, 503, 504)]
CHAIN_NSAMPLE = 2048


# more draws of the same experiment (other data seeds), gradients only, smaller samples: how the HIP / CPU spread ratio is
# DISTRIBUTED -- at B = 16 a handful of ReLU gates of the few-row bottleneck layers sit within rounding of zero, every fp32
# evaluation flips its own subset, and one seed is one draw
CHAIN_SEED_CASES = [(f"c2_b16_256_s{s}", 16, 256, s) for s in (501, 502, 503, 504)]


def gen_steps_chain(cases=CHAIN_CASES, nsample=CHAIN_NSAMPLE, kinds=("fwd", "bwd", "grad")):
    """The generator's backward CHAIN at the headline size, layer by layer (round-3 verdict, item 1): the oracle evaluated in
    fp64 -- every layer's output activation, the gradient entering every layer (d loss / d activation), the gradient of the
    generated batch (what the loss stack + discriminator hand the generator) and every parameter gradient, each as a strided
    sample (activations in NHWC order, parameters in logical OIHW order) -- next to the deviation of three fp32 evaluations
    of the same arithmetic from it ON THE SAME SAMPLE (oneDNN convolutions = what the reference runs, ATen's native
    convolutions, one thread).  A GPU test walks the chain from `final` back to `enc1` and reports where the HIP path's
    deviation from fp64 leaves the spread of the CPU fp32 evaluations (tests/test_hip_backward_chain.py)."""
    from oracle import terragan_oracle as Orc
    out = {}

    def sample_of(t, nhwc):
        t = t.detach()
        if nhwc and t.dim() == 4:
            t = t.permute(0, 2, 3, 1)
        t = t.reshape(-1)
        stride = max(1, t.numel() // nsample)
        return t[::stride][:nsample].double().clone(), stride

    def run(dtype, b, size, seed0, mode="step"):
        st = Orc.TrainState(0).to(dtype)
        real, mask = synth_batch(b, size, seed0)
        real, mask = real.to(dtype), mask.to(dtype)
        gk = Orc.trainable(st.gp)
        for k in gk:
            st.gp[k].requires_grad_(True)
        taps = {}
        gen = Orc.generator_forward(st.gp, real * mask, mask, True, taps)
        if mode == "gy":
            gy = torch.randn(gen.shape, generator=torch.Generator().manual_seed(5)).to(dtype) / gen.numel()
            total = (gen * gy).sum()
        else:
            g_loss, _ = Orc.inpainting_loss(st.vp, gen, real, mask, *st.w)
            total = g_loss + Orc.bce_logits(Orc.discriminator_forward(st.dp, gen, True), 1.0)
        names = list(taps)
        gs = torch.autograd.grad(total, [gen] + [taps[n] for n in names] + [st.gp[k] for k in gk])
        res = {"bwd/gen": (gs[0], True), "fwd/gen": (gen, True)}
        for n, g in zip(names, gs[1:1 + len(names)]):
            res[f"fwd/{n}"] = (taps[n], True)
            res[f"bwd/{n}"] = (g, True)
        for k, g in zip(gk, gs[1 + len(names):]):
            res[f"grad/{k}"] = (g, False)
        outd = {}
        for key, (t, nhwc) in res.items():
            if key.split("/")[0] not in kinds:
                continue
            smp, stride = sample_of(t, nhwc)
            td = t.detach().double()
            outd[key] = (smp, stride, float(td.abs().max()), float(td.pow(2).mean().sqrt()), t.numel())
        return outd

    for case in cases:
        tag, b, size, seed0 = case[:4]
        mode = case[4] if len(case) > 4 else "step"
        out[f"{tag}/cfg"] = np.array([b, size, seed0, nsample, 1 if mode == "gy" else 0], dtype=np.int64)
        r64 = run(torch.float64, b, size, seed0, mode)
        print(f"  {tag}: fp64 done", flush=True)
        for key, (smp, stride, mx, rms, n) in r64.items():
            out[f"{tag}/{key}/ref"] = smp.numpy()
            out[f"{tag}/{key}/stride"] = np.int64(stride)
            out[f"{tag}/{key}/scale"] = np.array([mx, rms, n], dtype=np.float64)      # max|ref|, rms(ref) over the FULL tensor, numel
        variants = [("onednn", True, None), ("native", False, None), ("onethread", True, 1)]
        out[f"{tag}/variants"] = np.array([v[0] for v in variants])
        devs = {key: [] for key in r64}
        for name, mkldnn_on, nthreads in variants:
            old_threads = torch.get_num_threads()
            if nthreads:
                torch.set_num_threads(nthreads)
            with torch.backends.mkldnn.flags(enabled=mkldnn_on):
                r32 = run(torch.float32, b, size, seed0, mode)
            torch.set_num_threads(old_threads)
            for key in r64:
                d = r32[key][0] - r64[key][0]
                devs[key].append([float(d.abs().max()), float(d.pow(2).mean().sqrt())])
            print(f"  {tag}: fp32 variant {name} done", flush=True)
        for key in r64:
            out[f"{tag}/{key}/dev"] = np.array(devs[key], dtype=np.float64)          # [variant][max|d|, rms(d)] on the sample
    return out



def gen_models_unc():
    """fp32-vs-fp64 deviation of the `models` fixtures' quantities (whole-generator forward/backward at tiny sizes, where
    BatchNorm runs over 2-3 values per channel at the bottleneck), from the oracle: [max|d|, sum|d|, ||d||_2, max|ref|] per
    tensor.  The GPU tests bound their error by the stated fp32 tolerance + K x this deviation, as the step tests do."""
    from oracle import terragan_oracle as Orc
    out = {}

    def dev4(a, c):
        a, c = a.double().flatten(), c.double().flatten()
        d = a - c
        return np.array([float(d.abs().max()), float(d.abs().sum()), float(d.norm()), float(a.abs().max())])

    for tag, b, h, w in MODEL_G_CASES:
        res = {}
        for dtype in (torch.float32, torch.float64):
            torch.manual_seed(7)
            gp = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in Orc.init_generator().items()}
            x, m = synth_batch(b, max(h, w), 300 + h)
            x, m = x[:, :, :h, :w].contiguous().to(dtype), m[:, :, :h, :w].contiguous().to(dtype)
            xm = (x * m).requires_grad_(True)
            keys = Orc.trainable(gp)
            for k in keys:
                gp[k].requires_grad_(True)
            y = Orc.generator_forward(gp, xm, m, True)
            gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(dtype)
            grads = torch.autograd.grad((y * gy).sum(), [xm] + [gp[k] for k in keys])
            bufs = {k: v.detach().clone() for k, v in gp.items() if "running" in k}
            with torch.no_grad():
                ye = Orc.generator_forward(gp, xm.detach(), m, False)
            res[dtype] = (y.detach(), grads, bufs, ye)
        (y32, g32, b32, e32), (y64, g64, b64, e64) = res[torch.float32], res[torch.float64]
        out[f"{tag}/out"] = dev4(y32, y64)
        out[f"{tag}/out_eval"] = dev4(e32, e64)
        out[f"{tag}/dx"] = dev4(g32[0], g64[0])
        for k, a, c in zip(keys, g32[1:], g64[1:]):
            out[f"{tag}/grad/{k}"] = dev4(a, c)
        for k in b32:
            out[f"{tag}/buf/{k}"] = dev4(b32[k], b64[k])
    return out


def gen_dataset():
    """The reference's InpaintingDataset (mvp_gan/src/utils/dataset.py:8-43) run on seeded PNG tiles.  The module imports
    `torchvision.transforms` at its top and never uses it: an empty stub module stands in.  The transform handed to it is
    what train.py:67-70 builds -- Resize((H,W)) + ToTensor() -- restated on PIL (torchvision's Resize on a PIL image IS
    PIL's bilinear resize, ToTensor is uint8/255 with a leading channel axis).  The fixture keeps the PNG bytes (inputs)
    and the tensors the reference dataset returned (expected outputs)."""
    import tempfile
    from PIL import Image
    tv = sys.modules.get("torchvision") or types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tv.transforms = tvt
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
    spec = importlib.util.spec_from_file_location("refpkg_dataset", os.path.join(REF, "mvp_gan/src/utils/dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def transform(size):
        def _tf(img):
            img = img.resize((size[1], size[0]), Image.BILINEAR)
            return torch.from_numpy(np.asarray(img, dtype=np.uint8).astype(np.float32) / 255.0).unsqueeze(0)
        return _tf

    out = {}
    rng = np.random.default_rng(31)
    with tempfile.TemporaryDirectory() as td:
        idir, mdir = os.path.join(td, "img"), os.path.join(td, "mask")
        os.makedirs(idir), os.makedirs(mdir)
        n = 5
        for i in range(n):
            h, w = [(40, 40), (37, 53), (64, 48), (40, 40), (50, 50)][i]
            img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
            # masks with values {0,255}, {0,1,2}, soft edges (anything > 0 after the resize is valid)
            if i % 3 == 0:
                msk = (rng.random((h, w)) > 0.4).astype(np.uint8) * 255
            elif i % 3 == 1:
                msk = rng.integers(0, 3, size=(h, w), dtype=np.uint8)
            else:
                msk = np.zeros((h, w), np.uint8)
                msk[h // 4: 3 * h // 4, w // 3:] = rng.integers(1, 256, size=(3 * h // 4 - h // 4, w - w // 3), dtype=np.uint8)
            for d, a in ((idir, img), (mdir, msk)):
                fn = os.path.join(d, f"tile_{i:02d}.png")
                Image.fromarray(a, mode="L").save(fn)
                out[f"png/{os.path.basename(d)}/{i}"] = np.frombuffer(open(fn, "rb").read(), dtype=np.uint8)
        out["n"] = np.int64(n)
        for tag, size in [("s32", (32, 32)), ("s48x40", (48, 40)), ("s64", (64, 64))]:
            ds = mod.InpaintingDataset(idir, mdir, transform=transform(size))
            assert len(ds) == n
            out[f"{tag}/size"] = np.array(size, dtype=np.int64)
            for i in range(n):
                item = ds[i]
                out[f"{tag}/image/{i}"] = item["image"].numpy()
                out[f"{tag}/mask/{i}"] = item["mask"].numpy().astype(np.uint8)
    return out


def main():
    only = sys.argv[1:]
    jobs = {"steps_unc": gen_steps_unc,           # oracle only: do not need the reference
            "steps_full_unc": lambda: gen_steps_unc(FULL_CASES, with_dp=False, alts=True),
            "models_unc": gen_models_unc,
            "steps_chain": gen_steps_chain,
            "steps_chain_seeds": lambda: gen_steps_chain(CHAIN_SEED_CASES, 512, ("bwd", "grad")),
            "steps_chain_gy": lambda: gen_steps_chain(CHAIN_GY_CASES, 256, ("bwd", "grad"))}
    if not only or any(n not in jobs for n in only):
        ref_pconv, ref_gen, ref_disc, ref_losses = _load_reference()
        jobs.update(_reference_jobs(ref_pconv, ref_gen, ref_disc, ref_losses))
    for name, fn in jobs.items():
        if only and name not in only:
            continue
        data = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **data)
        print(f"{name}: {len(data)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


def gen_metrics():
    """Logged quality metrics from the reference's own evaluation/metrics.py (calculate_boundary_quality :79-133 and
    MaskEvaluator._calculate_psnr/_calculate_ssim :47-76, the same arithmetic as utils/experiment_tracking.py:196-231).
    The module imports cv2 at the top for its OpenCV contour helper only; cv2 is absent here and is stubbed with an
    empty module (none of the functions used below touch it)."""
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("refpkg_metrics", os.path.join(REF, "mvp_gan/src/evaluation/metrics.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ev = mod.MaskEvaluator({})
    out = {}
    g = torch.Generator().manual_seed(21)
    cases = [("q48x40", 2, 48, 40, "blocks"), ("q64", 1, 64, 64, "holes30"), ("q33x70", 3, 33, 70, "border"),
             ("qones", 2, 32, 32, "ones"), ("qsame", 1, 40, 40, "blocks")]
    for tag, b, h, w, kind in cases:
        tgt = torch.rand(b, 1, h, w, generator=g)
        pred = tgt.clone() if tag == "qsame" else (tgt + 0.1 * torch.randn(b, 1, h, w, generator=g)).clamp(0, 1)
        m = mask_case(kind, b, h, w, g)
        out[f"{tag}/pred"], out[f"{tag}/target"], out[f"{tag}/mask"] = pred.numpy(), tgt.numpy(), m.numpy().astype(np.uint8)
        bq = mod.calculate_boundary_quality(pred, tgt, m)
        for k, v in bq.items():
            out[f"{tag}/{k}"] = np.float64(v)
        out[f"{tag}/psnr"] = np.float64(ev._calculate_psnr(pred, tgt))
        out[f"{tag}/ssim"] = np.float64(ev._calculate_ssim(pred, tgt))
        # utils/experiment_tracking.py:187-188 (that module needs mlflow/psutil/git to import; these two lines are plain torch)
        out[f"{tag}/l1_distance"] = np.float64(torch.nn.functional.l1_loss(pred, tgt).item())
        out[f"{tag}/l2_distance"] = np.float64(torch.nn.functional.mse_loss(pred, tgt, reduction="mean").sqrt().item())
    return out


def gen_validation(ref_gen, ref_disc, ref_losses):
    """Validation body of the reference loop (train.py:278-301) after one train step: G.eval(), D left in train mode."""
    out = {}
    bce = nn.BCEWithLogitsLoss()
    for tag, b, size in [("v2_128", 2, 128), ("v1_256", 1, 256)]:
        G, D, crit, oG, oD = _build(ref_gen, ref_disc, ref_losses)
        G.train(), D.train()
        real, mask = synth_batch(b, size, 70)
        _ref_step(G, D, crit, bce, oG, oD, real, mask)          # non-trivial running statistics and weights
        G.eval()
        vreal, vmask = synth_batch(b, size, 71)
        with torch.no_grad():
            gen = G(vreal * vmask, vmask)
            g_total = crit(gen, vreal, vmask)
            rv, fv = D(vreal), D(gen)
            d = 0.5 * (bce(rv, torch.ones_like(rv)).item() + bce(fv, torch.zeros_like(fv)).item())
        out[f"{tag}/cfg"] = np.array([b, size], dtype=np.int64)
        out[f"{tag}/val_g_loss"], out[f"{tag}/val_d_loss"] = np.float64(float(g_total)), np.float64(d)
        put(out, f"{tag}/gen", gen, full_limit=70000)
        for n, buf in D.named_buffers():
            if "running" in n:
                put(out, f"{tag}/dbuf/{n}", buf, full_limit=1024)
        out[f"{tag}/d_nbt"] = np.int64(int(D.model[3].num_batches_tracked))
    return out


def _reference_jobs(ref_pconv, ref_gen, ref_disc, ref_losses):
    return {
        "metrics": gen_metrics,
        "validation": lambda: gen_validation(ref_gen, ref_disc, ref_losses),
        "pconv_layers": lambda: gen_pconv(ref_pconv),
        "models": lambda: gen_models(ref_gen, ref_disc),
        "losses": lambda: gen_losses(ref_losses),
        "init": lambda: gen_init(ref_gen, ref_disc, ref_losses),
        "steps": lambda: gen_steps(ref_gen, ref_disc, ref_losses),
        "steps_dp8": lambda: gen_dp(ref_gen, ref_disc, ref_losses, [("dp8_128", 8, 4, 128)]),
        "steps_full": lambda: gen_steps_full(ref_gen, ref_disc, ref_losses),
        "dataset": gen_dataset,
    }


if __name__ == "__main__":
    main()
