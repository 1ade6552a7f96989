"""BASELINE config 3 (bf16 mixed precision): conv inner products on bf16 operands with fp32 accumulation, everything
else fp32.  Judged as SURVEY §8c says -- per-step losses within rtol 2e-2 of the fp32 path, not element-wise."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tg_hip import lib
    lib.load()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_precision():
    from tg_hip import ops as O
    yield
    O.set_precision("f32")


@pytest.mark.parametrize("shape", [(4, 64, 64, 64, 64, 3, 1, 1), (2, 32, 48, 256, 128, 3, 1, 1), (2, 64, 64, 128, 256, 4, 2, 1)])
def test_bf16_conv_close_to_f32(dev, shape):
    from tg_hip import ops as O
    B, H, W, Cin, Cout, k, s, p = shape
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).contiguous(memory_format=torch.channels_last).to(dev)
    bias = torch.randn(Cout, generator=g).to(dev)
    m = (torch.rand(B, H, W, generator=g) > 0.2).float().to(dev)
    _, ratio = O.mask_update(m, k, s, p)
    res = {}
    for prec in ("f32", "bf16"):
        O.set_precision(prec)
        y = O.conv_fwd(x, w, bias, k, s, p, in_mask=m, ratio=ratio)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        dx = O.conv_dgrad(dy, w, tuple(x.shape), k, s, p, in_mask=m)
        res[prec] = (y, dx)
    for a, b in zip(res["f32"], res["bf16"]):
        rel = float((a - b).double().norm() / a.double().norm())
        assert rel < 1e-2, rel                 # bf16 rounding of operands: ~2^-9 per product
    rel_dx = float((res["f32"][1] - res["bf16"][1]).double().norm() / res["f32"][1].double().norm())
    assert rel_dx > 1e-5, rel_dx               # every dgrad has a bf16 variant: the switch really changes the arithmetic


def test_bf16_train_step_losses(dev):
    from mvp_gan.src.models import Discriminator, PConvUNet
    from mvp_gan.src.train import train_step
    from mvp_gan.src.utils.losses import InpaintingLoss
    from oracle import terragan_oracle as Orc
    from tg_hip import ops as O
    out = {}
    for prec in ("f32", "bf16"):
        O.set_precision(prec)
        torch.manual_seed(0)
        G, D = PConvUNet(), Discriminator()
        crit = InpaintingLoss(0.1, 0.1, boundary_weight=0.5, device=torch.device("cpu"))
        G, D, crit = G.to(dev), D.to(dev), crit.to(dev)
        oG, oD = torch.optim.Adam(G.parameters(), lr=2e-4), torch.optim.Adam(D.parameters(), lr=2e-4)
        vals = []
        for s in range(3):
            real, mask = Orc.synth_batch(2, 256, 500 + s)
            o = train_step(G, D, crit, oG, oD, real.to(dev), mask.to(dev))
            vals.append((float(o["g_total"]), float(o["d_loss"])))
        out[prec] = vals
    for (g32, d32), (g16, d16) in zip(out["f32"], out["bf16"]):
        assert abs(g16 - g32) <= 2e-2 * abs(g32) and abs(d16 - d32) <= 2e-2 * abs(d32), (out["f32"], out["bf16"])
