"""Full-size (BASELINE configs[1]: 256x256, batch 16) checks through size-independent properties -- the CPU oracle
would need minutes here, so the kernels are checked against each other and against exact invariants:
  * adjointness: <conv(x), dy> == <x, dgrad(dy)> and == <W, wgrad(x, dy)> (fwd / dgrad / wgrad describe ONE bilinear form);
  * linearity of the conv in x;
  * mask path exactness (all-valid mask -> ratio 1 and mask 1 everywhere; hole pixels -> composite copies the input);
  * bitwise run-to-run determinism of a whole train step (fixed-order reductions, no float atomics).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tg_hip import lib
    lib.load()
    return torch.device("cuda:0")


def _dot(a, b):
    return float((a.double() * b.double()).sum())


# (name, B, H, W, Cin, Cout, k, s, p, masked): the layers that dominate the step + one of every kernel family
LAYERS = [("dec1", 16, 256, 256, 64, 64, 3, 1, 1, True), ("vgg2_2", 16, 128, 128, 128, 128, 3, 1, 1, False),
          ("enc2", 16, 128, 128, 64, 128, 5, 2, 2, True), ("d2", 16, 64, 64, 128, 256, 4, 2, 1, False),
          ("final", 16, 256, 256, 64, 1, 3, 1, 1, False), ("enc1", 16, 256, 256, 1, 64, 7, 2, 3, True),
          ("dec6", 16, 8, 8, 1024, 512, 3, 1, 1, True)]


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_adjoint_and_linearity(dev, layer):
    from tg_hip import ops as O
    _name, B, H, W, Cin, Cout, k, s, p, masked = layer
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    x2 = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).contiguous(memory_format=torch.channels_last).to(dev)
    m = (torch.rand(B, H, W, generator=g) > 0.25).float().to(dev) if masked else None
    y = O.conv_fwd(x, w, None, k, s, p, in_mask=m)
    dy = torch.randn(y.shape, generator=g).to(dev)
    dx = O.conv_dgrad(dy, w, tuple(x.shape), k, s, p, in_mask=m)
    dw, db = O.conv_wgrad(x, dy, w, k, s, p, in_mask=m)
    lhs = _dot(y, dy)
    scale = float(y.double().norm() * dy.double().norm()) + 1e-30
    assert abs(lhs - _dot(x, dx)) <= 2e-6 * scale, ("dgrad adjoint", lhs, _dot(x, dx))
    assert abs(lhs - _dot(w, dw)) <= 2e-6 * scale, ("wgrad adjoint", lhs, _dot(w, dw))
    assert torch.allclose(db.double(), dy.double().sum(dim=(0, 1, 2)), rtol=1e-5, atol=1e-3 * float(dy.numel() / Cout) ** 0.5 * 1e-2)
    # linearity: conv(2x - 3x2) == 2conv(x) - 3conv(x2)
    y2 = O.conv_fwd(x2, w, None, k, s, p, in_mask=m)
    xl = torch.empty_like(x)
    O._lib().tg_lincomb(O._p(x), 2.0, O._p(x2), -3.0, O._p(xl), x.numel(), O._stream())
    yl = O.conv_fwd(xl, w, None, k, s, p, in_mask=m)
    ref = 2.0 * y.double() - 3.0 * y2.double()
    assert float((yl.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-6


def test_mask_path_exact_and_composite(dev):
    from mvp_gan.src.models import PConvUNet
    from oracle import terragan_oracle as Orc
    from tg_hip import ops as O
    ones = torch.ones(16, 256, 256, device=dev)
    for k, s, p in [(7, 2, 3), (5, 2, 2), (3, 2, 1), (3, 1, 1)]:
        mo, ratio = O.mask_update(ones, k, s, p)
        assert bool((mo == 1).all())
        inner = ratio[:, 2:-2, 2:-2]
        assert bool((inner == 1).all()) and float(ratio.max()) <= k * k / ((k + 1) // 2) ** 2 + 1e-6    # borders renormalise up
    zeros = torch.zeros(16, 256, 256, device=dev)
    mo, ratio = O.mask_update(zeros, 3, 1, 1)
    assert bool((mo == 0).all()) and bool((ratio == 0).all())
    torch.manual_seed(0)
    G = PConvUNet().to(dev)
    real, mask = Orc.synth_batch(16, 256, 1000)
    real, mask = real.to(dev), mask.to(dev)
    with torch.no_grad():
        out = G(real * mask, mask)
    valid = mask > 0
    assert torch.equal(out[valid], (real * mask)[valid])        # generator.py:60-62: valid pixels are copied bit-exactly
    assert bool(((out >= 0) & (out <= 1)).all())


def test_train_step_bitwise_deterministic(dev):
    from mvp_gan.src.models import Discriminator, PConvUNet
    from mvp_gan.src.train import train_step
    from mvp_gan.src.utils.losses import InpaintingLoss
    from oracle import terragan_oracle as Orc
    real, mask = Orc.synth_batch(16, 256, 1001)
    real, mask = real.to(dev), mask.to(dev)
    snaps = []
    for _ in range(2):
        torch.manual_seed(0)
        G, D = PConvUNet(), Discriminator()
        crit = InpaintingLoss(0.1, 0.1, device=torch.device("cpu"))
        G, D, crit = G.to(dev), D.to(dev), crit.to(dev)
        oG, oD = torch.optim.Adam(G.parameters(), lr=2e-4), torch.optim.Adam(D.parameters(), lr=2e-4)
        for _s in range(2):
            out = train_step(G, D, crit, oG, oD, real, mask)
        snaps.append(([p_.detach().clone() for p_ in list(G.parameters()) + list(D.parameters())], out["gen"].clone(),
                      float(out["g_total"]), float(out["d_loss"])))
    assert snaps[0][2] == snaps[1][2] and snaps[0][3] == snaps[1][3]
    assert torch.equal(snaps[0][1], snaps[1][1])
    for a, b in zip(snaps[0][0], snaps[1][0]):
        assert torch.equal(a, b)
