"""Winograd paths of the stride-1 3x3 convolutions (F(2x2,3x3) fwd and dgrad, F(3x3,2x2) wgrad) against PyTorch-CPU fp64:
ragged tile edges, masks, bias/ratio/activation epilogue, fused activation-backward gate, accumulate, split-K."""
import numpy as np
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


# B, H, W, Cin, Cout, pad
CASES = [
    (2, 32, 32, 64, 64, 1),
    (2, 32, 48, 16, 32, 1),       # one 32-channel N tile
    (1, 40, 24, 8, 64, 1),        # ragged: 40 = 2.5 tiles, 24 = 1.5 tiles; one 8-channel K step
    (3, 17, 19, 24, 128, 1),      # odd sizes: the last Winograd tile is half outside
    (2, 16, 16, 1024, 128, 1),    # long K -> split-K slabs
    (1, 64, 48, 192, 64, 1),
    (2, 20, 20, 16, 64, 0),       # pad 0: output 18x18
    (1, 18, 22, 16, 64, 2),       # pad 2: output 20x24
    (8, 128, 128, 64, 64, 1),     # 512 work items on 256 workgroups: the staging pipeline runs across items (wino_pipe_kernel)
    (4, 96, 96, 64, 128, 1),      # 288 items: some workgroups walk two (different N tiles of one patch), most one
    (6, 80, 112, 24, 64, 1),      # 210 / 420 items of three 8-channel steps, ragged patches
]


def _ref_conv(x, w, bias, pad, mask, ratio, act):
    xin = x * mask[..., None] if mask is not None else x
    y = F.conv2d(xin.permute(0, 3, 1, 2).double(), w.double(), bias.double(), 1, pad).permute(0, 2, 3, 1)
    if ratio is not None:
        y = y * ratio[..., None].double()
    if act == "relu":
        y = y.clamp_min(0)
    return y


@pytest.mark.parametrize("case", CASES)
def test_wino_fwd(dev, case):
    from tg_hip import ops as O
    B, H, W, Cin, Cout, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    bias = torch.randn(Cout, generator=g) * 0.1
    mask = (torch.rand(B, H, W, generator=g) > 0.3).float()
    wd = w.contiguous(memory_format=torch.channels_last).to(dev)
    for use_mask, act in [(False, None), (True, "relu")]:
        m = mask if use_mask else None
        ratio = None
        if use_mask:
            _, ratio_d = O.mask_update(mask.to(dev), 3, 1, pad)
            ratio = ratio_d.cpu()
        ref = _ref_conv(x, w, bias, pad, m, ratio, act)
        y = O.conv_fwd(x.to(dev), wd, bias.to(dev), 3, 1, pad, in_mask=m.to(dev) if use_mask else None,
                       ratio=ratio_d if use_mask else None, act=O.ACT_RELU if act else O.ACT_NONE)
        err = (y.cpu().double() - ref).abs().max().item()
        assert err <= 2e-6 * max(1.0, ref.abs().max().item()) * (Cin / 64) ** 0.5 + 3e-6, (case, use_mask, err)


@pytest.mark.parametrize("case", CASES)
def test_wino_dgrad(dev, case):
    from tg_hip import ops as O
    B, H, W, Cin, Cout, pad = case
    if Cin % 64:                   # dgrad's output channels are Cin: the Winograd path needs a multiple of 64
        Cin, Cout = Cout, Cin if Cin % 8 == 0 else 8
    g = torch.Generator().manual_seed(sum(case) + 1)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    dy = torch.randn(B, Ho, Wo, Cout, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cout ** 0.5))
    mask = (torch.rand(B, H, W, generator=g) > 0.3).float()
    xact = torch.randn(B, H, W, Cin, generator=g)
    wd = w.contiguous(memory_format=torch.channels_last).to(dev)
    ref = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.double(), None, 1, pad).permute(0, 2, 3, 1)
    dx = O.conv_dgrad(dy.to(dev), wd, (B, H, W, Cin), 3, 1, pad)
    tol = 2e-6 * max(1.0, ref.abs().max().item()) * (Cout / 64) ** 0.5 + 3e-6
    assert (dx.cpu().double() - ref).abs().max().item() <= tol, case
    # masked + accumulate
    base = torch.randn(B, H, W, Cin, generator=g)
    out = base.clone().to(dev)
    O.conv_dgrad(dy.to(dev), wd, (B, H, W, Cin), 3, 1, pad, in_mask=mask.to(dev), out=out)
    ref2 = base.double() + ref * mask[..., None].double()
    assert (out.cpu().double() - ref2).abs().max().item() <= tol, case
    # fused LeakyReLU backward gate
    dxg = O.conv_dgrad(dy.to(dev), wd, (B, H, W, Cin), 3, 1, pad, gate=xact.to(dev), gate_act=O.ACT_LEAKY, gate_slope=0.2)
    ref3 = ref * torch.where(xact > 0, 1.0, 0.2).double()
    assert (dxg.cpu().double() - ref3).abs().max().item() <= tol, case


def test_wino_is_deterministic(dev):
    from tg_hip import ops as O
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 64, 64, 256, generator=g).to(dev)
    w = (torch.randn(128, 256, 3, 3, generator=g) * 0.02).contiguous(memory_format=torch.channels_last).to(dev)
    b = torch.zeros(128).to(dev)
    y1 = O.conv_fwd(x, w, b, 3, 1, 1)
    y2 = O.conv_fwd(x, w, b, 3, 1, 1)
    assert torch.equal(y1, y2)


def test_cu_reserve_does_not_change_the_bits(dev):
    """tg_set_cu_reserve (data-parallel runs: CUs left free for RCCL) sizes the GRID of the persistent Winograd kernels only;
    the split-K plans -- the summation order -- are made for 256 CUs whatever the reserve.  Forward, dgrad and wgrad of
    split-K and multi-item layers, F(2x2,3x3) and F(2x2,2x2), must be bitwise equal with and without a reserve."""
    from tg_hip import lib as L
    from tg_hip import ops as O
    lib = L.load()
    g = torch.Generator().manual_seed(9)
    res = {}
    cases = [(2, 32, 32, 768, 256, 3, 1, 1), (8, 128, 128, 64, 64, 3, 1, 1), (4, 64, 64, 128, 256, 4, 2, 1), (16, 16, 16, 1024, 512, 3, 1, 1)]
    data = []
    for (B, H, W, Cin, Cout, k, s_, p_) in cases:
        x = torch.randn(B, H, W, Cin, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.02).contiguous(memory_format=torch.channels_last).to(dev)
        Ho = (H + 2 * p_ - k) // s_ + 1
        dy = torch.randn(B, Ho, Ho, Cout, generator=g).to(dev)
        data.append((x, w, dy))
    try:
        for r in (0, 8, 40):
            L.check(lib.tg_set_cu_reserve(r), "tg_set_cu_reserve")
            out = []
            for (B, H, W, Cin, Cout, k, s_, p_), (x, w, dy) in zip(cases, data):
                out.append(O.conv_fwd(x, w, None, k, s_, p_))
                out.append(O.conv_dgrad(dy, w, tuple(x.shape), k, s_, p_))
                out.append(O.conv_wgrad(x, dy, w, k, s_, p_, want_bias=False)[0].contiguous())
            res[r] = out
    finally:
        L.check(lib.tg_set_cu_reserve(0), "tg_set_cu_reserve")
    for r in (8, 40):
        for a, b_ in zip(res[0], res[r]):
            assert torch.equal(a, b_), r


def test_work_stealing_does_not_change_the_bits(dev):
    """tg_set_work_stealing (data-parallel runs): the persistent Winograd workgroups pull their items from per-XCD queues
    instead of walking a fixed list.  Which workgroup computes an item never changes a result: forward / dgrad of layers with
    one, a few and many items per workgroup, split-K, the cross-item pipeline, the gated variant and the discriminator's
    F(2x2,2x2) kernels must be bitwise equal in both modes -- also launched back to back (the counters reset themselves) and
    twice in a row."""
    from tg_hip import lib as L
    from tg_hip import ops as O
    lib = L.load()
    g = torch.Generator().manual_seed(11)
    cases = [(2, 32, 32, 768, 256, 3, 1, 1), (8, 128, 128, 64, 64, 3, 1, 1), (4, 96, 96, 64, 128, 3, 1, 1), (4, 64, 64, 128, 256, 4, 2, 1),
             (16, 16, 16, 1024, 512, 3, 1, 1), (6, 80, 112, 24, 64, 3, 1, 1), (3, 17, 19, 24, 128, 3, 1, 1)]
    data = []
    for (B, H, W, Cin, Cout, k, s_, p_) in cases:
        x = torch.randn(B, H, W, Cin, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.02).contiguous(memory_format=torch.channels_last).to(dev)
        Ho, Wo = (H + 2 * p_ - k) // s_ + 1, (W + 2 * p_ - k) // s_ + 1
        dy = torch.randn(B, Ho, Wo, Cout, generator=g).to(dev)
        gate = torch.randn(B, H, W, Cin, generator=g).to(dev)
        data.append((x, w, dy, gate))
    res = {}
    try:
        for mode in (0, 2, 1, 0):          # 2: every launch pulls from the queues; 1: those with two or more items per workgroup
            L.check(lib.tg_set_work_stealing(mode), "tg_set_work_stealing")
            out = []
            for (B, H, W, Cin, Cout, k, s_, p_), (x, w, dy, gate) in zip(cases, data):
                out.append(O.conv_fwd(x, w, None, k, s_, p_, act=O.ACT_RELU))
                out.append(O.conv_dgrad(dy, w, tuple(x.shape), k, s_, p_))
                if Cin % 64 == 0:
                    out.append(O.conv_dgrad(dy, w, tuple(x.shape), k, s_, p_, gate=gate, gate_act=O.ACT_LEAKY, gate_slope=0.2))
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(out)
    finally:
        L.check(lib.tg_set_work_stealing(0), "tg_set_work_stealing")
    ref = res[0][0]
    for outs in (res[2][0], res[1][0], res[0][1]):
        for a, b_ in zip(ref, outs):
            assert torch.equal(a, b_)


def test_epilogue_stores_under_repetition(dev):
    """The persistent kernel's epilogue stores through buffer descriptors.  A store with a scalar-register offset let a later
    vector write to its data registers change the stored value on gfx950 -- a few dozen wrong elements per launch, different
    ones from run to run.  Forward (bias + ReLU), accumulate and gated dgrad, six launches each on a multi-item
    and a ragged geometry: every launch against fp64 and bitwise equal to the first."""
    from tg_hip import ops as O
    g = torch.Generator().manual_seed(77)
    for (B, H, W, Cin, Cout) in [(4, 64, 64, 64, 64), (2, 40, 56, 128, 64)]:
        x = torch.randn(B, H, W, Cin, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
        bias = torch.randn(Cout, generator=g) * 0.1
        dy = torch.randn(B, H, W, Cout, generator=g)
        xact = torch.randn(B, H, W, Cin, generator=g)
        base = torch.randn(B, H, W, Cin, generator=g)
        wd = w.contiguous(memory_format=torch.channels_last).to(dev)
        xd, bd, dyd, xad = x.to(dev), bias.to(dev), dy.to(dev), xact.to(dev)
        ref_f = _ref_conv(x, w, bias, 1, None, None, "relu")
        ref_d = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
        ref_g = ref_d * torch.where(xact > 0, 1.0, 0.2).double()
        ref_a = base.double() + ref_d
        tol_f = 3e-6 * max(1.0, ref_f.abs().max().item()) * (Cin / 64) ** 0.5 + 3e-6
        tol_d = 3e-6 * max(1.0, ref_d.abs().max().item()) * (Cout / 64) ** 0.5 + 3e-6
        first = None
        for rep in range(6):
            y = O.conv_fwd(xd, wd, bd, 3, 1, 1, act=O.ACT_RELU)
            dg = O.conv_dgrad(dyd, wd, (B, H, W, Cin), 3, 1, 1, gate=xad, gate_act=O.ACT_LEAKY, gate_slope=0.2)
            acc = base.clone().to(dev)
            O.conv_dgrad(dyd, wd, (B, H, W, Cin), 3, 1, 1, out=acc)
            assert (y.cpu().double() - ref_f).abs().max().item() <= tol_f, (rep, "fwd")
            assert (dg.cpu().double() - ref_g).abs().max().item() <= tol_d, (rep, "gated dgrad")
            assert (acc.cpu().double() - ref_a).abs().max().item() <= tol_d, (rep, "accumulate")
            if first is None:
                first = (y, dg, acc)
            else:
                assert torch.equal(y, first[0]) and torch.equal(dg, first[1]) and torch.equal(acc, first[2]), rep


def test_outputs_of_2gb_and_more_leave_the_descriptor_epilogue(dev):
    """The persistent kernel's epilogue addresses dst through a buffer descriptor (32-bit byte offsets): a launch whose OUTPUT
    reaches 2 GB (input well below) must take the kernels with 64-bit store addresses.  8 x 1024 x 1024, 8 -> 64 channels:
    2.1 GB of output in one launch against the same convolution in two batch halves (1.07 GB each, descriptor epilogue)."""
    from tg_hip import ops as O
    g = torch.Generator().manual_seed(2031)
    x = torch.randn(8, 1024, 1024, 8, generator=g).to(dev)
    w = (torch.randn(64, 8, 3, 3, generator=g) / 8.5).contiguous(memory_format=torch.channels_last).to(dev)
    b = (torch.randn(64, generator=g) * 0.1).to(dev)
    y = O.conv_fwd(x, w, b, 3, 1, 1, act=O.ACT_RELU)
    assert y.numel() * 4 >= 1 << 31
    for h in range(2):
        yh = O.conv_fwd(x[4 * h:4 * h + 4].contiguous(), w, b, 3, 1, 1, act=O.ACT_RELU)
        d = (y[4 * h:4 * h + 4] - yh).abs().max().item()
        assert d <= 2e-6 * max(1.0, yh.abs().max().item()), (h, d)
        del yh
    # the last pixels of the last image (the far end of the 2.1 GB) against fp64
    ref = F.conv2d(x[7:8, 1000:].permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu(), 1, 1).permute(0, 2, 3, 1).clamp_min(0)
    got = y[7:8, 1001:].double().cpu()
    assert (got - ref[:, 1:]).abs().max().item() <= 3e-6 * max(1.0, ref.abs().max().item()) + 3e-6


# B, H, W, Cin, Cout, pad
WGRAD_CASES = [
    (2, 32, 32, 64, 64, 1),
    (1, 40, 24, 64, 128, 1),      # ragged strips: 24 = 1.5 strips of 16 pixels
    (3, 17, 19, 128, 64, 1),      # odd sizes: half tiles at the right / bottom edge
    (2, 20, 20, 64, 64, 0),       # pad 0
    (1, 18, 22, 64, 64, 2),       # pad 2
    (4, 64, 64, 192, 64, 1),      # several Cin tiles, split-K over strips
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wino_wgrad(dev, case):
    from tg_hip import ops as O
    B, H, W, Cin, Cout, pad = case
    g = torch.Generator().manual_seed(sum(case) + 2)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    x = torch.randn(B, H, W, Cin, generator=g)
    dy = torch.randn(B, Ho, Wo, Cout, generator=g)
    w = torch.zeros(Cout, Cin, 3, 3).contiguous(memory_format=torch.channels_last).to(dev)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, 3, 3), dy.permute(0, 3, 1, 2).double(),
                                      stride=1, padding=pad)
    dw, db = O.conv_wgrad(x.to(dev), dy.to(dev), w, 3, 1, pad)
    err = (dw.cpu().double() - ref).abs().max().item()
    assert err <= 3e-6 * ref.abs().max().item() + 1e-5, (case, err, ref.abs().max().item())
    assert torch.allclose(db.cpu().double(), dy.double().sum((0, 1, 2)), atol=1e-3, rtol=1e-5)
    dw2, _ = O.conv_wgrad(x.to(dev), dy.to(dev), w, 3, 1, pad)
    assert torch.equal(dw, dw2)          # deterministic split-K reduction


def test_winograd_kernels_are_the_ones_that_run(dev, tmp_path):
    """Eligible stride-1 3x3 layers must go through wino_kernel (launch tag 4064) / wino_wgrad_kernel (4164): a silent
    fall-back to the direct kernels would keep every parity test green and lose the speed."""
    import csv
    import ctypes as C
    from tg_hip import lib as L
    from tg_hip import ops as O
    lib = L.load()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 32, 32, 64, generator=g).to(dev)
    w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).contiguous(memory_format=torch.channels_last).to(dev)
    b = torch.zeros(64).to(dev)
    for kind in (0, 1, 2, 3):          # drop records of earlier tests
        lib.tg_prof_summary(kind, None, None, None, None)
    lib.tg_prof_enable(1)
    y = O.conv_fwd(x, w, b, 3, 1, 1)
    O.conv_dgrad(y, w, tuple(x.shape), 3, 1, 1)
    O.conv_wgrad(x, y, w, 3, 1, 1)
    torch.cuda.synchronize()
    lib.tg_prof_enable(0)
    path = str(tmp_path / "launches.csv")
    assert lib.tg_prof_dump(path.encode()) == 0
    tags = [(r["kind"], r["cfg"]) for r in csv.DictReader(open(path))]
    for kind in (0, 1, 2, 3):
        lib.tg_prof_summary(kind, None, None, None, None)
    assert tags.count(("0", "4064")) == 2 and ("1", "4164") in tags, tags


# 5x5 / stride 2 / pad 2 layers (enc2, enc3) run as a 3x3 stride-1 Winograd convolution over the space-to-depth input
S2D_CASES = [(2, 64, 64, 64, 128), (1, 64, 96, 16, 64), (2, 64, 64, 128, 256)]   # B, H, W, Cin, Cout


@pytest.mark.parametrize("case", S2D_CASES)
def test_s2d_5x5_stride2(dev, case):
    from tg_hip import ops as O
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 5, 5, generator=g) / (5 * Cin ** 0.5)
    bias = torch.randn(Cout, generator=g) * 0.1
    mask = (torch.rand(B, H, W, generator=g) > 0.3).float()
    wd = w.contiguous(memory_format=torch.channels_last).to(dev)
    mo, ratio_d = O.mask_update(mask.to(dev), 5, 2, 2)
    xm = (x * mask[..., None]).permute(0, 3, 1, 2).double()
    ref = F.conv2d(xm, w.double(), bias.double(), 2, 2).permute(0, 2, 3, 1) * ratio_d.cpu()[..., None].double()
    y = O.conv_fwd(x.to(dev), wd, bias.to(dev), 5, 2, 2, in_mask=mask.to(dev), ratio=ratio_d)
    tol = 3e-6 * max(1.0, ref.abs().max().item()) * (Cin / 16) ** 0.5 + 3e-6
    assert (y.cpu().double() - ref).abs().max().item() <= tol, case
    # dgrad (masked, accumulate) and wgrad (masked input)
    dy = torch.randn(B, H // 2, W // 2, Cout, generator=g)
    refdx = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.double(), None, 2, 2, output_padding=1).permute(0, 2, 3, 1)
    base = torch.randn(B, H, W, Cin, generator=g)
    out = base.clone().to(dev)
    O.conv_dgrad(dy.to(dev), wd, (B, H, W, Cin), 5, 2, 2, in_mask=mask.to(dev), out=out)
    ref2 = base.double() + refdx * mask[..., None].double()
    assert (out.cpu().double() - ref2).abs().max().item() <= 3e-6 * max(1.0, refdx.abs().max().item()) * (Cout / 16) ** 0.5 + 3e-6, case
    refdw = torch.nn.grad.conv2d_weight(xm, (Cout, Cin, 5, 5), dy.permute(0, 3, 1, 2).double(), stride=2, padding=2)
    dw, db = O.conv_wgrad(x.to(dev), dy.to(dev), wd, 5, 2, 2, in_mask=mask.to(dev))
    assert (dw.cpu().double() - refdw).abs().max().item() <= 3e-6 * refdw.abs().max().item() + 1e-5, case
    assert torch.allclose(db.cpu().double(), dy.double().sum((0, 1, 2)), atol=1e-3, rtol=1e-5)


# ---- F(2x2,2x2): the discriminator's 4x4 / stride-2 / pad-1 convolutions (wino22.inc) ---------------------------------------
# B, H, W, Cin, Cout
W22_CASES = [
    (2, 32, 32, 64, 128),
    (1, 64, 32, 8, 64),          # one 8-channel step per parity phase
    (3, 36, 44, 24, 64),         # ragged: 18 x 22 outputs = 1.1 x 1.4 patches
    (2, 32, 32, 256, 64),        # K = 1024 -> split-K slabs
    (1, 128, 128, 64, 128),      # the d2 geometry at batch 1
    (4, 32, 32, 128, 256),
    (12, 128, 128, 64, 128),     # 384 (forward) / 1536 (dgrad) work items on 256 persistent workgroups
]


@pytest.mark.parametrize("case", W22_CASES)
def test_wino22_fwd(dev, case):
    from tg_hip import ops as O
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) / (4 * Cin ** 0.5)
    bias = torch.randn(Cout, generator=g) * 0.1
    mask = (torch.rand(B, H, W, generator=g) > 0.3).float()
    wd = w.contiguous(memory_format=torch.channels_last).to(dev)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), 2, 1).permute(0, 2, 3, 1)
    y = O.conv_fwd(x.to(dev), wd, bias.to(dev), 4, 2, 1)
    tol = 2e-6 * max(1.0, ref.abs().max().item()) * (Cin / 16) ** 0.5 + 3e-6
    assert (y.cpu().double() - ref).abs().max().item() <= tol, case
    # masked input + LeakyReLU epilogue
    refm = F.conv2d((x * mask[..., None]).permute(0, 3, 1, 2).double(), w.double(), bias.double(), 2, 1).permute(0, 2, 3, 1)
    refm = torch.where(refm > 0, refm, 0.2 * refm)
    ym = O.conv_fwd(x.to(dev), wd, bias.to(dev), 4, 2, 1, in_mask=mask.to(dev), act=O.ACT_LEAKY, slope=0.2)
    assert (ym.cpu().double() - refm).abs().max().item() <= tol, case
    assert torch.equal(y, O.conv_fwd(x.to(dev), wd, bias.to(dev), 4, 2, 1))


@pytest.mark.parametrize("case", W22_CASES)
def test_wino22_dgrad(dev, case):
    from tg_hip import ops as O
    B, H, W, Cout, Cin = case          # swapped: dgrad's output channels (Cin) must be a multiple of 64
    g = torch.Generator().manual_seed(sum(case) + 1)
    Ho, Wo = H // 2, W // 2
    dy = torch.randn(B, Ho, Wo, Cout, generator=g)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) / (4 * Cout ** 0.5)
    mask = (torch.rand(B, H, W, generator=g) > 0.3).float()
    xact = torch.randn(B, H, W, Cin, generator=g)
    wd = w.contiguous(memory_format=torch.channels_last).to(dev)
    ref = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.double(), None, 2, 1).permute(0, 2, 3, 1)
    dx = O.conv_dgrad(dy.to(dev), wd, (B, H, W, Cin), 4, 2, 1)
    tol = 2e-6 * max(1.0, ref.abs().max().item()) * (Cout / 16) ** 0.5 + 3e-6
    assert (dx.cpu().double() - ref).abs().max().item() <= tol, case
    base = torch.randn(B, H, W, Cin, generator=g)
    out = base.clone().to(dev)
    O.conv_dgrad(dy.to(dev), wd, (B, H, W, Cin), 4, 2, 1, in_mask=mask.to(dev), out=out)
    ref2 = base.double() + ref * mask[..., None].double()
    assert (out.cpu().double() - ref2).abs().max().item() <= tol, case
    dxg = O.conv_dgrad(dy.to(dev), wd, (B, H, W, Cin), 4, 2, 1, gate=xact.to(dev), gate_act=O.ACT_LEAKY, gate_slope=0.2)
    ref3 = ref * torch.where(xact > 0, 1.0, 0.2).double()
    assert (dxg.cpu().double() - ref3).abs().max().item() <= tol, case
    assert torch.equal(dx, O.conv_dgrad(dy.to(dev), wd, (B, H, W, Cin), 4, 2, 1))


def test_wino22_is_the_kernel_that_runs(dev, tmp_path):
    """The discriminator's 64->128 layer must go through wino22_kernel (launch tag 4022), forward and dgrad, and its
    weight gradient through wino22_wgrad_kernel (4122)."""
    import csv
    from tg_hip import lib as L, ops as O
    lib = L.load()
    x = torch.randn(2, 64, 64, 64, device=dev)
    w = (torch.randn(128, 64, 4, 4, device=dev) * 0.03).contiguous(memory_format=torch.channels_last)
    b = torch.zeros(128, device=dev)
    for kind in (0, 1, 2, 3):          # drop records of earlier tests
        lib.tg_prof_summary(kind, None, None, None, None)
    lib.tg_prof_enable(1)
    y = O.conv_fwd(x, w, b, 4, 2, 1)
    O.conv_dgrad(y, w, (2, 64, 64, 64), 4, 2, 1)
    O.conv_wgrad(x, y, w, 4, 2, 1)
    torch.cuda.synchronize()
    lib.tg_prof_enable(0)
    path = str(tmp_path / "launches.csv")
    assert lib.tg_prof_dump(path.encode()) == 0
    tags = [(r["kind"], r["cfg"]) for r in csv.DictReader(open(path))]
    for kind in (0, 1, 2, 3):
        lib.tg_prof_summary(kind, None, None, None, None)
    assert tags.count(("0", "4022")) == 2 and ("1", "4122") in tags, tags


# B, H, W, Cin, Cout
W22_WGRAD_CASES = [
    (2, 32, 32, 64, 64),
    (1, 64, 32, 64, 128),
    (3, 36, 44, 128, 64),        # ragged strips: 22 outputs = 1.4 strips; 18 rows = 9 tile rows
    (4, 64, 64, 64, 128),        # split-K over strips
    (2, 32, 32, 256, 128),
]


@pytest.mark.parametrize("case", W22_WGRAD_CASES)
def test_wino22_wgrad(dev, case):
    from tg_hip import ops as O
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case) + 2)
    x = torch.randn(B, H, W, Cin, generator=g)
    dy = torch.randn(B, H // 2, W // 2, Cout, generator=g)
    w = torch.zeros(Cout, Cin, 4, 4).contiguous(memory_format=torch.channels_last).to(dev)
    ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double(), (Cout, Cin, 4, 4), dy.permute(0, 3, 1, 2).double(),
                                      stride=2, padding=1)
    dw, db = O.conv_wgrad(x.to(dev), dy.to(dev), w, 4, 2, 1)
    err = (dw.cpu().double() - ref).abs().max().item()
    assert err <= 3e-6 * ref.abs().max().item() + 1e-5, (case, err, ref.abs().max().item())
    assert torch.allclose(db.cpu().double(), dy.double().sum((0, 1, 2)), atol=1e-3, rtol=1e-5)
    dw2, _ = O.conv_wgrad(x.to(dev), dy.to(dev), w, 4, 2, 1)
    assert torch.equal(dw, dw2)          # deterministic split-K reduction


# ---- Winograd F(4x4,3x3) (csrc/wino44.inc: the frozen VGG trunk's forward and dgrad; requested per call with wino4=True) ----------
# B, H, W, Cin, Cout, pad
W44_CASES = [
    (2, 32, 32, 64, 64, 1),       # two 16 x 32 blocks per image, 8 K steps
    (1, 16, 32, 16, 64, 1),       # one block, a single trip of two K steps
    (2, 40, 72, 16, 128, 1),      # ragged: 2.5 x 2.25 blocks, two N tiles
    (3, 19, 37, 48, 64, 1),       # odd sizes: the last tiles hang over the edge
    (1, 64, 64, 256, 256, 1),     # VGG conv3_x shape family: 32 K steps, 4 N tiles
    (2, 18, 34, 16, 64, 0),       # pad 0: 16 x 32 outputs
    (9, 64, 96, 64, 64, 1),       # 108 blocks... several work items per workgroup only with > 256: see next
    (20, 64, 128, 32, 64, 1),     # 320 items on 256 persistent workgroups
]


@pytest.mark.parametrize("case", W44_CASES)
def test_wino44_fwd_dgrad(dev, case):
    """F(4x4,3x3) against fp64: forward with bias + ReLU, dgrad plain / accumulate / fused ReLU-backward gate.  Error budget:
    several times the F(2x2,3x3) kernel's (transform coefficients up to 8 and 1/24: rms 6-7x, maximum ~20x on random data,
    tools/conv_fuzz.py)."""
    from tg_hip import ops as O
    B, H, W, Cin, Cout, pad = case
    g = torch.Generator().manual_seed(sum(case) + 44)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    bias = torch.randn(Cout, generator=g) * 0.1
    wd = w.contiguous(memory_format=torch.channels_last).to(dev)
    ref = _ref_conv(x, w, bias, pad, None, None, "relu")
    y = O.conv_fwd(x.to(dev), wd, bias.to(dev), 3, 1, pad, act=O.ACT_RELU, wino4=True)
    err = (y.cpu().double() - ref).abs().max().item()
    assert err <= 8e-6 * max(1.0, ref.abs().max().item()) * (Cin / 64) ** 0.5 + 1e-5, (case, "fwd", err)
    # the F(2x2,3x3) result of the same call differs (the request really changes the kernel) but agrees to fp32 accuracy
    y2 = O.conv_fwd(x.to(dev), wd, bias.to(dev), 3, 1, pad, act=O.ACT_RELU)
    assert not torch.equal(y, y2)
    # dgrad: N = Cin must be a multiple of 64 on this path, K = Cout a multiple of 8
    Ci2, Co2 = (Cin, Cout) if Cin % 64 == 0 else (Cout, Cin)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    dy = torch.randn(B, Ho, Wo, Co2, generator=g)
    w2 = (torch.randn(Co2, Ci2, 3, 3, generator=g) / (3 * Co2 ** 0.5))
    w2d = w2.contiguous(memory_format=torch.channels_last).to(dev)
    xact = torch.randn(B, H, W, Ci2, generator=g)
    refd = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w2.double(), None, 1, pad).permute(0, 2, 3, 1)
    tol = 8e-6 * max(1.0, refd.abs().max().item()) * (Co2 / 64) ** 0.5 + 1e-5
    dx = O.conv_dgrad(dy.to(dev), w2d, (B, H, W, Ci2), 3, 1, pad, wino4=True)
    assert (dx.cpu().double() - refd).abs().max().item() <= tol, (case, "dgrad")
    base = torch.randn(B, H, W, Ci2, generator=g)
    out = base.clone().to(dev)
    O.conv_dgrad(dy.to(dev), w2d, (B, H, W, Ci2), 3, 1, pad, out=out, wino4=True)
    assert (out.cpu().double() - (base.double() + refd)).abs().max().item() <= tol, (case, "accumulate")
    dxg = O.conv_dgrad(dy.to(dev), w2d, (B, H, W, Ci2), 3, 1, pad, gate=xact.to(dev), gate_act=O.ACT_RELU, wino4=True)
    assert (dxg.cpu().double() - refd * (xact > 0).double()).abs().max().item() <= tol, (case, "gate")
    # bitwise reproducible
    assert torch.equal(dx, O.conv_dgrad(dy.to(dev), w2d, (B, H, W, Ci2), 3, 1, pad, wino4=True))


def test_wino44_prepared_weights_meet_a_launch_that_cannot_use_them(dev):
    """The layout of a PREPARED weight buffer (F(4x4) or F(2x2) image) follows the geometry alone -- what the caller's cache key
    holds -- never the launch: one cached weight then serves launches that run on wino44_kernel and launches that cannot
    (row scales, an input mask, a batch whose output reaches 2 GB), in either order.  The latter transform F(2x2) weights
    into their workspace (wino44_prepared_unusable, csrc/wino44.inc)."""
    from tg_hip import ops as O
    B, H, W, Cin, Cout = 2, 32, 64, 64, 128
    g = torch.Generator().manual_seed(4402)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    bias = torch.randn(Cout, generator=g) * 0.1
    mask = (torch.rand(B, H, W, generator=g) > 0.3).float()
    _, ratio = O.mask_update(mask.to(dev), 3, 1, 1)
    ref_plain = _ref_conv(x, w, bias, 1, None, None, None)
    ref_pc = _ref_conv(x, w, bias, 1, mask, ratio.cpu(), None)
    tol = lambda r: 8e-6 * max(1.0, r.abs().max().item()) + 1e-5
    dy = torch.randn(B, H, W, Cout, generator=g)
    refd = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
    for order in ("plain_first", "masked_first"):
        wd = w.contiguous(memory_format=torch.channels_last).to(dev)      # a fresh tensor: a fresh cache entry
        calls = [("plain", None, None, ref_plain), ("pc", mask.to(dev), ratio, ref_pc)]
        for name, m, r, ref in (calls if order == "plain_first" else calls[::-1]) * 2:
            y = O.conv_fwd(x.to(dev), wd, bias.to(dev), 3, 1, 1, in_mask=m, ratio=r, wino4=True)
            err = (y.cpu().double() - ref).abs().max().item()
            assert err <= tol(ref), (order, name, err)
        dcalls = [("plain", None, refd), ("masked", mask.to(dev), refd * mask[..., None].double())]
        for name, m, ref in (dcalls if order == "plain_first" else dcalls[::-1]) * 2:
            dx = O.conv_dgrad(dy.to(dev), wd, (B, H, W, Cin), 3, 1, 1, in_mask=m, wino4=True)
            err = (dx.cpu().double() - ref).abs().max().item()
            assert err <= tol(ref), (order, "dgrad", name, err)
    # the batch-size limit: 128 x 128 x 128 channels, B = 4 (wino44_kernel) and B = 256 (2 GB of output: wino_kernel) on ONE weight
    g2 = torch.Generator().manual_seed(4403)
    w2 = (torch.randn(128, 128, 3, 3, generator=g2) / 34.0).contiguous(memory_format=torch.channels_last).to(dev)
    b2 = (torch.randn(128, generator=g2) * 0.1).to(dev)
    xs = torch.randn(4, 128, 128, 128, generator=g2).to(dev)
    ys = O.conv_fwd(xs, w2, b2, 3, 1, 1, wino4=True)
    xb = xs.repeat(64, 1, 1, 1)
    yb = O.conv_fwd(xb, w2, b2, 3, 1, 1, wino4=True)
    assert yb.numel() * 4 >= 1 << 31
    ys2 = O.conv_fwd(xs, w2, b2, 3, 1, 1, wino4=True)
    assert torch.equal(ys, ys2)
    for i in (0, 17, 63):
        d = (yb[4 * i:4 * i + 4] - ys).abs().max().item()
        assert d <= 2e-5 * max(1.0, ys.abs().max().item()), (i, d)


# B, H, W, Cin, Cout, k, stride, pad -- tg_conv_fwd_pool: conv -> ReLU -> 2x2 max-pool in one call
POOL_CASES = [
    (8, 128, 128, 64, 64, 3, 1, 1),     # the cross-item pipeline, two items per workgroup: the pooled tensor leaves the output transform
    (4, 96, 96, 64, 128, 3, 1, 1),      # two N tiles
    (6, 80, 112, 24, 64, 3, 1, 1),      # ragged: 80 = 5 tiles, 112 = 7 tiles, three K steps
    (3, 40, 24, 16, 64, 3, 1, 1),       # tiles half outside (40 = 2.5, 24 = 1.5): pooled rows / columns beyond the edge are dropped
    (2, 20, 20, 16, 64, 3, 1, 0),       # pad 0: 18 x 18 output
    (2, 16, 16, 1024, 128, 3, 1, 1),    # split-K: the pool kernel runs on y
    (2, 32, 32, 64, 64, 4, 2, 1),       # F(2x2,2x2) layer: the pool kernel runs on y
    (2, 32, 32, 1, 64, 3, 1, 1),        # 1-channel source (VGG conv1_1's kernel family)
]


@pytest.mark.parametrize("case", POOL_CASES)
def test_conv_fwd_pool_equals_conv_then_pool(dev, case):
    """The pooled tensor of tg_conv_fwd_pool and the convolution output next to it are bitwise what tg_conv_fwd_p followed by
    tg_maxpool2_fwd give (a maximum rounds nothing) -- with the fixed item lists and with the work-stealing queues of DP runs."""
    from tg_hip import lib as L
    from tg_hip import ops as O
    lib = L.load()
    B, H, W, Cin, Cout, k, s_, p_ = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (k * Cin ** 0.5)).contiguous(memory_format=torch.channels_last).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    y0 = O.conv_fwd(x, w, b, k, s_, p_, act=O.ACT_RELU)
    p0 = O.maxpool2_fwd(y0)
    try:
        for mode in (0, 2, 0):
            L.check(lib.tg_set_work_stealing(mode), "tg_set_work_stealing")
            y1, p1 = O.conv_fwd(x, w, b, k, s_, p_, act=O.ACT_RELU, pool=True)
            assert p1.shape == p0.shape
            assert torch.equal(y1, y0) and torch.equal(p1, p0), (mode, float((p1 - p0).abs().max()))
    finally:
        L.check(lib.tg_set_work_stealing(0), "tg_set_work_stealing")


def test_conv_fwd_pool_is_fused_on_the_trunk_layers(dev, tmp_path):
    """VGG conv1_2 / conv2_2 shapes: the pooled tensor is written by the Winograd launch itself (tag 4064) -- seen in its launch
    record, whose algorithmic bytes then include the pooled tensor."""
    import csv
    from tg_hip import lib as L
    from tg_hip import ops as O
    lib = L.load()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 64, 64, 64, generator=g).to(dev)
    w = (torch.randn(128, 64, 3, 3, generator=g) * 0.05).contiguous(memory_format=torch.channels_last).to(dev)
    b = torch.zeros(128).to(dev)
    O.conv_fwd(x, w, b, 3, 1, 1, act=O.ACT_RELU, pool=True)          # (weights prepared outside the recorded region)
    for kind in (0, 1, 2, 3):
        lib.tg_prof_summary(kind, None, None, None, None)
    lib.tg_prof_enable(1)
    y, yp = O.conv_fwd(x, w, b, 3, 1, 1, act=O.ACT_RELU, pool=True)
    torch.cuda.synchronize()
    lib.tg_prof_enable(0)
    path = str(tmp_path / "launches.csv")
    assert lib.tg_prof_dump(path.encode()) == 0
    rows = list(csv.DictReader(open(path)))
    for kind in (0, 1, 2, 3):
        lib.tg_prof_summary(kind, None, None, None, None)
    assert [(r["kind"], r["cfg"]) for r in rows] == [("0", "4064")], rows
    plain_mb = 4 * (x.numel() + y.numel() + w.numel() + y.numel() // 128) / 1e6
    assert abs(float(rows[0]["alg_mb"]) - (plain_mb + 4 * yp.numel() / 1e6)) < 1e-2 * plain_mb, (rows[0]["alg_mb"], plain_mb)
    assert torch.equal(yp, O.maxpool2_fwd(y))
    assert float(yp.abs().sum()) > 0


@pytest.mark.parametrize("case", [(32, 128, 128, 64, 64), (16, 96, 96, 64, 128), (40, 48, 80, 24, 64)])
def test_conv_fwd_pool_code_and_its_backward(dev, case):
    """tg_conv_fwd_pool_code: the pooled tensor equals conv + ReLU + max-pool bit for bit, and tg_maxpool2_bwd_code applied to the
    code reproduces tg_maxpool2_bwd(relu_gate) on the full-resolution activation -- which this path never writes."""
    from tg_hip import ops as O
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    x[:, : H // 4] = 0                                   # a band of exact zeros: windows whose maximum is 0 (gate closed, ties)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).contiguous(memory_format=torch.channels_last).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    b[: Cout // 2] = -0.5                                # ... and channels that are negative there before the ReLU
    assert O.conv_pool_code_supported(tuple(x.shape), Cout)
    y = O.conv_fwd(x, w, b, 3, 1, 1, act=O.ACT_RELU)
    p0 = O.maxpool2_fwd(y)
    p1, code = O.conv_fwd_pool_code(x, w, b)
    assert torch.equal(p1, p0)
    dp = torch.randn(p0.shape, generator=g).to(dev)
    nb = B // 2
    dx0 = O.maxpool2_bwd(dp[:nb].contiguous(), y[:nb], relu_gate=True)
    dx1 = O.maxpool2_bwd_code(dp[:nb].contiguous(), code)
    assert torch.equal(dx1, dx0), float((dx1 - dx0).abs().max())
    assert float(dx1.abs().sum()) > 0 and float((dx1 == 0).float().mean()) > 0.75
    assert not O.conv_pool_code_supported((2, 32, 32, 128), 128)          # a K-split plan: stays on the two-output form
