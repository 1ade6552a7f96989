"""Op-level parity of the C ABI against plain PyTorch-CPU fp32 references at awkward shapes: channel counts that are not
multiples of 4 (scalar fallbacks), odd spatial sizes, strides/kernels outside the network's own set, tiny and empty-ish
masks.  These paths are not exercised by the train step itself."""
import math

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


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close(a, b, rtol=1e-4, atol=1e-5):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= atol + rtol * b.abs().max().item(), f"max err {err:.3e} (ref max {b.abs().max().item():.3e})"


CONVS = [  # B, H, W, Cin, Cout, k, s, p
    (2, 13, 17, 3, 5, 3, 1, 1),       # odd everything, scalar K path, N < 32
    (2, 16, 16, 6, 34, 3, 2, 1),      # C % 4 != 0, N not a tile multiple
    (1, 9, 31, 8, 96, 5, 2, 2),       # N = 96 (64-tile + guard)
    (3, 20, 12, 32, 64, 1, 1, 0),     # 1x1 conv
    (2, 24, 40, 64, 64, 3, 1, 1),     # patch kernel, OW not a multiple of 16
    (2, 17, 33, 128, 128, 3, 1, 1),   # patch kernel 128-config with overhanging tiles
    (2, 32, 32, 64, 192, 4, 2, 1),    # stride-2 k4 (merged parity classes in dgrad)
    (1, 8, 8, 512, 512, 3, 1, 1),     # small spatial, split-K
    (2, 10, 10, 16, 16, 3, 3, 0),     # stride 3 (9 dgrad classes, not merged)
    (2, 16, 32, 1, 64, 3, 1, 1),      # 1 -> 64 (c1conv)
    (2, 16, 32, 64, 1, 3, 1, 1),      # 64 -> 1 (to1conv64)
    (2, 12, 20, 1, 128, 4, 2, 1),     # 1 -> 128, two channel groups
    (2, 32, 40, 1, 64, 4, 2, 1),      # 1 -> 64 (D conv0): its dgrad = four 2x2-tap classes from one LDS patch (ragged: 16 x 20 class grid)
    (3, 16, 32, 1, 64, 4, 2, 1),      # the same with whole tiles
    (3, 16, 16, 512, 1, 4, 1, 1),     # 512 -> 1, 4x4 (the discriminator's last conv: to1convw / to1wgradw)
    (2, 9, 13, 256, 1, 3, 1, 1),      # 256 -> 1, 3x3, odd sizes
]


@pytest.mark.parametrize("cfg", CONVS, ids=[f"{c[3]}to{c[4]}_k{c[5]}s{c[6]}_{c[1]}x{c[2]}" for c in CONVS])
def test_conv_fwd_dgrad_wgrad(dev, cfg):
    from tg_hip import ops as O
    B, H, W, Cin, Cout, k, s, p = cfg
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    m = (torch.rand(B, 1, H, W, generator=g) > 0.3).float()
    ones = torch.ones(1, 1, k, k)
    ssum = F.conv2d(m, ones, None, s, p)
    ratio = (k * k) / (ssum + 1e-8) * (ssum > 0).float()
    y_ref = F.leaky_relu((F.conv2d(x * m, w, b, s, p)) * ratio, 0.2)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)

    xd, md = nhwc(x.detach()).to(dev), m[:, 0].contiguous().to(dev)
    wd = w.detach().contiguous(memory_format=torch.channels_last).to(dev)
    mo, rd = O.mask_update(md, k, s, p)
    assert torch.equal(mo.cpu(), (ssum[:, 0] > 0).float()) and torch.equal(rd.cpu(), ratio[:, 0])       # integer mask path: exact
    y = O.conv_fwd(xd, wd, b.detach().to(dev), k, s, p, in_mask=md, ratio=rd, act=O.ACT_LEAKY, slope=0.2)
    close(nchw(y), y_ref)
    # backward: dy_conv = gy * leaky'(y) * ratio
    dyr = O.act_bwd(nhwc(gy).to(dev), y, O.ACT_LEAKY, 0.2, ratio=rd, inplace=False)
    dx = O.conv_dgrad(dyr, wd, tuple(xd.shape), k, s, p, in_mask=md)
    close(nchw(dx), x.grad, rtol=2e-4)
    dw, db = O.conv_wgrad(xd, dyr, wd, k, s, p, in_mask=md)
    close(dw, w.grad, rtol=2e-4, atol=1e-4)
    close(db, b.grad, rtol=2e-4, atol=1e-4)
    # accumulate + gated variants
    base = torch.randn(xd.shape, generator=g).to(dev)
    acc = O.conv_dgrad(dyr, wd, tuple(xd.shape), k, s, p, in_mask=md, out=base.clone())
    close(acc, base + dx, rtol=2e-4)
    gate = torch.randn(xd.shape, generator=g).to(dev)
    gated = O.conv_dgrad(dyr, wd, tuple(xd.shape), k, s, p, in_mask=md, gate=gate, gate_act=O.ACT_RELU)
    close(gated, dx * (gate > 0).float(), rtol=2e-4)


@pytest.mark.parametrize("C,rows", [(1, 50), (3, 1000), (6, 77), (64, 4096), (192, 333), (512, 64), (1024, 7)])
def test_batchnorm_fwd_bwd(dev, C, rows):
    from tg_hip import ops as O
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(rows, C, generator=g) * 2 + 5).requires_grad_(True)     # mean >> 0: exercises the shifted sums
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(C, generator=g, requires_grad=True)
    rm, rv = torch.zeros(C), torch.ones(C)
    ratio = torch.rand(rows, generator=g) + 0.5
    y = F.relu(F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5))
    gy = torch.randn(rows, C, generator=g)
    xin = x                                                   # y = BN(x); conv bias gradient = sum_rows ratio * dx_bn
    y.backward(gy)
    xd = x.detach().reshape(1, 1, rows, C).contiguous().to(dev)
    rmd, rvd, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    mean, rstd = O.bn_stats(xd, rmd, rvd, nbt)
    close(rmd, rm, rtol=1e-5, atol=1e-6)
    close(rvd, rv, rtol=1e-5, atol=1e-6)
    assert int(nbt) == 1
    a = O.bn_act_fwd(xd, mean, rstd, gamma.detach().to(dev), beta.detach().to(dev), O.ACT_RELU)
    close(a.reshape(rows, C), y)
    dy, dgamma, dbeta, dbias = O.bn_act_bwd(gy.reshape(1, 1, rows, C).contiguous().to(dev), xd, mean, rstd, gamma.detach().to(dev),
                                            beta.detach().to(dev), O.ACT_RELU, ratio=ratio.to(dev), inplace=False)
    close(dy.reshape(rows, C), x.grad * ratio[:, None], rtol=2e-4, atol=1e-5)
    close(dgamma, gamma.grad, rtol=2e-4, atol=1e-4)
    close(dbeta, beta.grad, rtol=2e-4, atol=1e-4)
    close(dbias, (x.grad * ratio[:, None]).sum(0), rtol=1e-3, atol=2e-3 * math.sqrt(rows))


@pytest.mark.parametrize("shape", [(2, 5, 7, 8, 4), (1, 4, 4, 6, 3), (2, 9, 6, 64, 64), (2, 3, 5, 4, 0), (1, 1, 1, 5, 2)])
@pytest.mark.parametrize("delta", [(0, 0), (1, 0), (-1, 1), (3, 2)])
def test_upcat_fwd_bwd(dev, shape, delta):
    """bilinear x2 + _pad_to_match (incl. negative = crop, generator.py:78-84) + concat + merged-mask multiply."""
    from tg_hip import ops as O
    B, h, w, Cu, Cs = shape
    H, W = 2 * h + delta[0], 2 * w + delta[1]
    if H < 1 or W < 1:
        pytest.skip("empty")
    g = torch.Generator().manual_seed(1)
    up = torch.randn(B, Cu, h, w, generator=g, requires_grad=True)
    skip = torch.randn(B, Cs, H, W, generator=g, requires_grad=True) if Cs else None
    om = (torch.rand(B, 1, H, W, generator=g) > 0.3).float()
    u = F.interpolate(up, scale_factor=2, mode="bilinear", align_corners=False)
    dY, dX = H - u.shape[2], W - u.shape[3]
    u = F.pad(u, [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2])
    ref = (torch.cat([u, skip], 1) if Cs else u) * om
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    out = O.upcat_fwd(nhwc(up.detach()).to(dev), nhwc(skip.detach()).to(dev) if Cs else None, H, W, out_mask=om[:, 0].contiguous().to(dev))
    close(nchw(out), ref)
    dup, dskip = O.upcat_bwd(nhwc(gy * om).to(dev), h, w, Cu)
    close(nchw(dup), up.grad, rtol=2e-4)
    if Cs:
        close(nchw(dskip), skip.grad)
    # nearest-up mask merge uses the same padding rule
    um = (torch.rand(B, 1, h, w, generator=g) > 0.5).float()
    sm = (torch.rand(B, 1, H, W, generator=g) > 0.5).float()
    un = F.interpolate(um, scale_factor=2, mode="nearest")
    un = F.pad(un, [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2])
    mm = O.mask_up_merge(um[:, 0].contiguous().to(dev), sm[:, 0].contiguous().to(dev))
    assert torch.equal(mm.cpu(), torch.maximum(un, sm)[:, 0])


@pytest.mark.parametrize("rows_g,C,groups", [(4096, 512, 2), (16384, 256, 2), (2100, 128, 3), (65536, 128, 2)])
def test_grouped_batchnorm_equals_per_pass_calls(dev, rows_g, C, groups):
    """tg_bn_fwd_grouped / tg_bn_act_bwd_grouped / tg_bn_running_update_multi (the stacked D(fake) / D(real) passes of the train
    step: statistics per pass, one set of launches) against the per-pass calls they replace, bit for bit: outputs, statistics,
    the in-place input gradient, parameter gradients accumulated in pass order, running statistics replayed in a given order."""
    from tg_hip import ops as O
    g = torch.Generator().manual_seed(rows_g + C)
    y = (torch.randn(groups * rows_g, C, generator=g) * 2 + torch.randn(C, generator=g)).reshape(groups, rows_g, 1, C).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
    da = torch.randn(groups, rows_g, 1, C, generator=g).to(dev)
    mean2, rstd2, a2 = O.bn_fwd_grouped(y, groups, gamma, beta, O.ACT_LEAKY, 0.2)
    d2 = da.clone()
    _dy, dg2, db2, dbi2 = O.bn_act_bwd_grouped(d2, y, groups, mean2, rstd2, gamma, beta, O.ACT_LEAKY, 0.2)
    order = [0, 1, 0] if groups == 2 else [2, 0, 1, 1]
    rm2, rv2, nbt2 = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    O.bn_running_update_multi(mean2, rstd2, rows_g, order, rm2, rv2, nbt2)
    # the per-pass calls
    rm1, rv1, nbt1 = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    stats, dg1 = [], None
    for gi in range(groups):
        mean, rstd = O.bn_stats(y[gi])
        a1 = O.bn_act_fwd(y[gi], mean, rstd, gamma, beta, O.ACT_LEAKY, 0.2)
        assert torch.equal(mean, mean2[gi]) and torch.equal(rstd, rstd2[gi]) and torch.equal(a1, a2[gi])
        d1 = da[gi].clone()
        _d, g1, b1, bi1 = O.bn_act_bwd(d1, y[gi], mean, rstd, gamma, beta, O.ACT_LEAKY, 0.2)
        assert torch.equal(d1, d2[gi])
        if gi == 0:
            dg1, db1, dbi1 = g1, b1, bi1
        else:
            O.axpby_(g1, 1.0, 1.0, dg1), O.axpby_(b1, 1.0, 1.0, db1), O.axpby_(bi1, 1.0, 1.0, dbi1)
        stats.append((mean, rstd))
    assert torch.equal(dg1, dg2) and torch.equal(db1, db2) and torch.equal(dbi1, dbi2)
    for gi in order:
        O.bn_running_update(stats[gi][0], stats[gi][1], rows_g, rm1, rv1, nbt1)
    assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2) and int(nbt1) == int(nbt2) == len(order)


@pytest.mark.parametrize("shape", [(2, 5, 7, 8, 4), (2, 9, 6, 64, 64), (2, 3, 5, 4, 0), (1, 1, 1, 8, 4), (3, 16, 16, 128, 64), (1, 1, 6, 4, 4)])
def test_upcat_x2_equals_generic(dev, shape):
    """The exact x2 kernels (one thread per SOURCE pixel: upcat_fwd_x2_kernel / upcat_bwd_x2_kernel, skip half in the same
    launch) against the generic per-output-element gather on the same inputs: same interpolation expression, same operand
    order -- equal bit for bit, forward, upsample adjoint and skip slice.  TG_NO_UPCAT_X2 is read per call."""
    import os
    from tg_hip import ops as O
    B, h, w, Cu, Cs = shape
    g = torch.Generator().manual_seed(11)
    up = torch.randn(B, h, w, Cu, generator=g).to(dev)
    skip = torch.randn(B, 2 * h, 2 * w, Cs, generator=g).to(dev) if Cs else None
    om = (torch.rand(B, 2 * h, 2 * w, generator=g) > 0.3).float().to(dev)
    dcat = torch.randn(B, 2 * h, 2 * w, Cu + Cs, generator=g).to(dev)
    res = {}
    for mode in ("x2", "generic"):
        if mode == "generic":
            os.environ["TG_NO_UPCAT_X2"] = "1"
        try:
            out = O.upcat_fwd(up, skip, 2 * h, 2 * w, out_mask=om)
            out_nm = O.upcat_fwd(up, skip, 2 * h, 2 * w)
            dup, dskip = O.upcat_bwd(dcat, h, w, Cu)
        finally:
            os.environ.pop("TG_NO_UPCAT_X2", None)
        res[mode] = (out, out_nm, dup, dskip)
    for a, b in zip(res["x2"], res["generic"]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(2, 8, 8, 4), (1, 7, 9, 3), (2, 5, 6, 64), (1, 2, 2, 1)])
def test_maxpool(dev, shape):
    from tg_hip import ops as O
    B, H, W, C = shape
    g = torch.Generator().manual_seed(3)
    x = F.relu(torch.randn(B, C, H, W, generator=g)).requires_grad_(True)       # post-ReLU input, ties at 0
    y = F.max_pool2d(x, 2, 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = nhwc(x.detach()).to(dev)
    close(nchw(O.maxpool2_fwd(xd)), y, rtol=0, atol=0)
    dx = O.maxpool2_bwd(nhwc(gy).to(dev), xd)
    # ties among zeros may pick another zero than ATen; ReLU's backward kills those anyway
    close(nchw(dx) * (x.detach() > 0).float().to(dev), x.grad * (x.detach() > 0).float(), rtol=0, atol=0)
    dxg = O.maxpool2_bwd(nhwc(gy).to(dev), xd, relu_gate=True)
    close(nchw(dxg), x.grad * (x.detach() > 0).float(), rtol=0, atol=0)


@pytest.mark.parametrize("shape", [(1, 2, 2), (2, 33, 17), (3, 64, 64)])
def test_pixel_losses_and_bce_shapes(dev, shape):
    from oracle import terragan_oracle as Orc
    from tg_hip import ops as O
    B, H, W = shape
    g = torch.Generator().manual_seed(5)
    pred = torch.rand(B, 1, H, W, generator=g, requires_grad=True)
    tgt = torch.rand(B, 1, H, W, generator=g)
    for kind in ("rand", "ones", "zeros"):
        m = {"rand": (torch.rand(B, 1, H, W, generator=g) > 0.4).float(), "ones": torch.ones(B, 1, H, W), "zeros": torch.zeros(B, 1, H, W)}[kind]
        pred.grad = None
        tot = (pred - tgt).abs().mean() + 0.1 * Orc.tv_loss(pred * (1 - m)) + 0.5 * Orc.boundary_loss(pred, tgt, m)
        tot.backward()
        out5, dp = O.pixel_losses(pred.detach()[:, 0].contiguous().to(dev), tgt[:, 0].contiguous().to(dev), m[:, 0].contiguous().to(dev),
                                  1.0, 0.1, 0.5)
        close(out5[4], tot.detach(), rtol=1e-5, atol=1e-7)
        close(dp, pred.grad[:, 0], rtol=1e-4, atol=1e-9)
    z = torch.randn(B, H, W, generator=g) * 4
    for t in (0.0, 1.0):
        zz = z.clone().requires_grad_(True)
        l = F.binary_cross_entropy_with_logits(zz, torch.full_like(zz, t))
        l.backward()
        lo, dz = O.bce_logits(z.to(dev), t)
        close(lo[0], l.detach(), rtol=1e-5, atol=1e-7)
        close(dz, zz.grad, rtol=1e-4, atol=1e-9)


def test_adam_matches_torch(dev):
    from mvp_gan.src.train import hip_adam_step
    g = torch.Generator().manual_seed(9)
    shapes = [(7,), (3, 5), (64, 3, 3, 3), (130, 1, 1, 1), (1,)]
    p_ref = [torch.randn(s_, generator=g).requires_grad_(True) for s_ in shapes]
    p_hip = [p.detach().clone().to(dev).requires_grad_(True) for p in p_ref]
    o_ref, o_hip = torch.optim.Adam(p_ref, lr=3e-3), torch.optim.Adam(p_hip, lr=3e-3)
    for step in range(4):
        for pr, ph in zip(p_ref, p_hip):
            gr = torch.randn(pr.shape, generator=g)
            pr.grad, ph.grad = gr.clone(), gr.clone().to(dev)
        o_ref.step()
        hip_adam_step(o_hip)
    for pr, ph in zip(p_ref, p_hip):
        close(ph, pr, rtol=1e-6, atol=1e-7)
        close(o_hip.state[ph]["exp_avg_sq"], o_ref.state[pr]["exp_avg_sq"], rtol=1e-6, atol=1e-12)
    sd = o_hip.state_dict()                               # still a torch.optim.Adam state dict
    torch.optim.Adam([torch.zeros(s_) for s_ in shapes], lr=3e-3).load_state_dict(
        {"state": {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in sd["state"].items()},
         "param_groups": sd["param_groups"]})


@pytest.mark.parametrize("shape", [(2, 256, 256), (3, 72, 40), (1, 64, 64), (2, 100, 36)])
def test_mask_pyramid_equals_per_level_ops(dev, shape):
    """tg_mask_pyramid (all 21 mask ops of a generator forward in one launch) is bit-identical to the per-level
    tg_mask_update / tg_mask_up_merge calls (pconv.py:33-40, generator.py:51-54,68-74), odd sizes included."""
    from tg_hip import engine as E
    from tg_hip import ops as O
    B, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    mask = (torch.rand(B, H, W, generator=g) > 0.6).float()
    mask[:, : H // 3, : W // 2] = 0.0                       # a large hole: zero window sums deep into the pyramid
    mask = mask.to(dev)
    enc = [(k, s, p) for (_n, _ci, _co, k, s, p) in E.G_ENC]
    dec = [(k, s, p) for (_n, _ci, _co, k, s, p) in E.G_DEC]
    m, er, dmasks, dr = O.mask_pyramid(mask, enc, dec)
    rm, rer = [mask], [None]
    for (k, s, p) in enc:
        mo, r = O.mask_update(rm[-1], k, s, p)
        rm.append(mo), rer.append(r)
    dm = rm[7]
    for i, (k, s, p) in enumerate(dec):
        mm = O.mask_up_merge(dm, rm[6 - i] if i < 6 else mask)
        dm, r = O.mask_update(mm, k, s, p)
        assert torch.equal(dmasks[i], mm) and torch.equal(dr[i], r), i
    for i in range(1, 8):
        assert torch.equal(m[i], rm[i]) and torch.equal(er[i], rer[i]), i


@pytest.mark.parametrize("rows,C,act,with_ratio", [(4, 512, 1, True), (1024, 512, 1, True), (2048, 256, 2, False), (300, 64, 1, True),
                                                   (37, 128, 0, False)])
def test_small_batchnorm_one_launch_forms(dev, rows, C, act, with_ratio):
    """Few-row maps take the one-launch BatchNorm kernels (bn_fwd_small_kernel / bn_bwd_small_kernel).  Against fp64:
    statistics, running update, output, and the backward's dy / dgamma / dbeta / closed-form conv-bias gradient."""
    from tg_hip import ops as O
    g = torch.Generator().manual_seed(rows + C)
    y = (torch.randn(rows, C, generator=g) * 2.0 + 3.0)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rm0, rv0 = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    dout = torch.randn(rows, C, generator=g)
    ratio = (torch.rand(rows, generator=g) * 3.0) if with_ratio else None
    slope = 0.2
    yd = y.to(dev).reshape(1, rows, 1, C)
    rm, rv, nbt = rm0.clone().to(dev), rv0.clone().to(dev), torch.zeros((), dtype=torch.int64, device=dev)
    mean, rstd, out = O.bn_fwd(yd, gamma.to(dev), beta.to(dev), act, slope, rm, rv, nbt)
    y64 = y.double()
    mu, var = y64.mean(0), y64.var(0, unbiased=False)
    close(mean, mu, 1e-6, 1e-6)
    close(rstd, 1.0 / torch.sqrt(var + 1e-5), 2e-6, 0)
    close(rm, 0.9 * rm0.double() + 0.1 * mu, 1e-6, 1e-7)
    close(rv, 0.9 * rv0.double() + 0.1 * var * rows / (rows - 1), 2e-6, 1e-7)
    assert int(nbt) == 1
    xh = (y64 - mu) / torch.sqrt(var + 1e-5)
    z = xh * gamma.double() + beta.double()
    ref_out = z.clamp_min(0) if act == 1 else (torch.where(z > 0, z, slope * z) if act == 2 else z)
    close(out.reshape(rows, C), ref_out, 1e-5, 1e-5)
    # backward
    gate = (z > 0).double() if act == 1 else (torch.where(z > 0, 1.0, slope).double() if act == 2 else torch.ones_like(z))
    gg = dout.double() * gate
    dbeta, dgamma = gg.sum(0), (gg * xh).sum(0)
    dy = gamma.double() / torch.sqrt(var + 1e-5) * (gg - dbeta / rows - xh * dgamma / rows)
    if ratio is not None:
        dy = dy * ratio.double()[:, None]
    dyd, dg, db_, dbias = O.bn_act_bwd(dout.to(dev).reshape(1, rows, 1, C), yd, mean, rstd, gamma.to(dev), beta.to(dev), act, slope,
                                       ratio=None if ratio is None else ratio.to(dev).reshape(1, rows, 1), inplace=False)
    close(dg, dgamma, 2e-5, 2e-5)
    close(db_, dbeta, 2e-5, 2e-5)
    close(dyd.reshape(rows, C), dy, 5e-5, 1e-5)
    close(dbias, dy.sum(0), 1e-4, 1e-4 * float(dy.abs().sum(0).max()))


@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 40, 24), (1, 20, 52), (2, 4, 16)])
def test_conv_bn_on_load_equals_the_two_call_form(dev, shape):
    """tg_conv_fwd_bnin / tg_conv_wgrad_bnin (`final` over dec1's ReLU(BN(y)), the activation never written) against
    tg_bn_act_fwd + tg_conv_fwd / tg_conv_wgrad: the staging applies the same rounding sequence, so the results are the same bits;
    geometries without the kernel say so through tg_conv_bnin_supported."""
    from tg_hip import ops as O
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    y = (torch.randn(B, H, W, 64, generator=g) * 2 + 0.5).to(dev)
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
    mean, rstd = O.bn_stats(y)
    w = (torch.randn(1, 64, 3, 3, generator=g) * 0.1).contiguous(memory_format=torch.channels_last).to(dev)
    b = torch.randn(1, generator=g).to(dev)
    dz = torch.randn(B, H, W, 1, generator=g).to(dev)
    assert O.conv_bnin_supported(tuple(y.shape), 1, 3, 1, 1) and O.conv_bnin_supported(tuple(y.shape), 1, 3, 1, 1, wgrad=True)
    a = O.bn_act_fwd(y, mean, rstd, gamma, beta, O.ACT_RELU)
    bn = (mean, rstd, gamma, beta, O.ACT_RELU, 0.0)
    z0 = O.conv_fwd(a, w, b, 3, 1, 1)
    z1 = O.conv_fwd_bnin(y, bn, w, b, 3, 1, 1)
    assert torch.equal(z0, z1), float((z0 - z1).abs().max())
    dw0, db0 = O.conv_wgrad(a, dz, w, 3, 1, 1)
    dw1, db1 = O.conv_wgrad(y, dz, w, 3, 1, 1, in_bn=bn)
    assert torch.equal(dw0, dw1) and torch.equal(db0, db1), float((dw0 - dw1).abs().max())
    # and against fp64
    ad = torch.relu((y.double().cpu() - mean.double().cpu()) * rstd.double().cpu() * gamma.double().cpu() + beta.double().cpu())
    zr = F.conv2d(nchw(ad), w.double().cpu(), b.double().cpu(), 1, 1)
    close(nchw(z1), zr, rtol=1e-4, atol=1e-4)
    for bad in [(B, H, W, 128), (B, H, W + 2, 64) if W % 4 == 0 else (B, 2, W, 64)]:
        if bad[2] % 4 or bad[3] != 64 or bad[1] < 4:
            assert not O.conv_bnin_supported(bad, 1, 3, 1, 1)
    assert not O.conv_bnin_supported(tuple(y.shape), 64, 3, 1, 1)


def test_generator_with_and_without_bn_on_load(dev):
    """The generator step with the decoder's BatchNorm + ReLU applied on load by the consumers (`final` for dec1, the next level's
    upsample + concat for dec2 ... dec5: the default) and with the activations written and read back (TG_NO_BNIN=1 /
    engine.BNIN_FINAL = BNIN_UPCAT = False): outputs, every gradient and the running statistics bit for bit equal."""
    from mvp_gan.src.models.generator import PConvUNet
    from tg_hip import engine as E
    torch.manual_seed(3)
    res = []
    G0 = PConvUNet()
    sd = {k: v.clone() for k, v in G0.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.rand(3, 128, 128, generator=g).to(dev)
    m = (torch.rand(3, 128, 128, generator=g) > 0.3).float().to(dev)
    dout = torch.randn(3, 128, 128, generator=g).to(dev)
    old = E.BNIN_FINAL, E.BNIN_UPCAT, E.BN_CONV1
    E.BN_CONV1 = False          # (the recomputed-dgrad BatchNorm backward has its own test: it is equal to rounding, not bit for bit)
    try:
        for flag in (True, False):
            E.BNIN_FINAL = E.BNIN_UPCAT = flag          # (dec2 and dec3 have more than 2048 rows here: deferred to their upsample)
            G = PConvUNet()
            G.load_state_dict(sd)
            G = G.to(dev)
            P = G._tensors()
            out, ctx = E.generator_forward(P, x * m, m, training=True)
            assert ctx.bnin_final == flag
            grads, _ = E.generator_backward(P, ctx, dout.clone())
            res.append((out.clone(), {k: v.clone() for k, v in grads.items()}, {k: v.clone() for k, v in G.named_buffers()}))
    finally:
        E.BNIN_FINAL, E.BNIN_UPCAT, E.BN_CONV1 = old
    (o1, g1, b1), (o0, g0, b0) = res
    assert torch.equal(o1, o0)
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k
    for k in b0:
        assert torch.equal(b1[k], b0[k]), k


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 32), (1, 9, 13, 128, 0), (3, 8, 20, 32, 64)])
def test_upcat_bn_on_load_equals_the_two_call_form(dev, shape):
    """tg_upcat_fwd_bn (upsample + concat over ReLU(BN(up)) formed on load) against tg_bn_act_fwd + tg_upcat_fwd: same bits."""
    from tg_hip import ops as O
    B, h, w, Cu, Cs = shape
    g = torch.Generator().manual_seed(sum(shape))
    y = (torch.randn(B, h, w, Cu, generator=g) * 1.5 - 0.2).to(dev)
    gamma, beta = (torch.rand(Cu, generator=g) + 0.5).to(dev), (torch.randn(Cu, generator=g) * 0.3).to(dev)
    mean, rstd = O.bn_stats(y)
    skip = torch.randn(B, 2 * h, 2 * w, Cs, generator=g).to(dev) if Cs else None
    om = (torch.rand(B, 2 * h, 2 * w, generator=g) > 0.3).float().to(dev)
    assert O.upcat_bn_supported(tuple(y.shape), None if skip is None else tuple(skip.shape), 2 * h, 2 * w)
    assert not O.upcat_bn_supported(tuple(y.shape), None if skip is None else tuple(skip.shape), 2 * h + 1, 2 * w)
    a = O.bn_act_fwd(y, mean, rstd, gamma, beta, O.ACT_RELU)
    c0 = O.upcat_fwd(a, skip, 2 * h, 2 * w, out_mask=om)
    c1 = O.upcat_fwd(y, skip, 2 * h, 2 * w, out_mask=om, up_bn=(mean, rstd, gamma, beta, O.ACT_RELU, 0.0))
    assert torch.equal(c0, c1), float((c0 - c1).abs().max())


@pytest.mark.parametrize("shape", [(2, 64, 64, 64), (3, 40, 24, 32), (2, 20, 52, 128), (2, 17, 23, 16)])
def test_bn_backward_over_a_recomputed_1channel_dgrad(dev, shape):
    """tg_bn_act_bwd_conv1 (dec1's BatchNorm backward with `final`'s input gradient recomputed from the 1-channel dz in both passes)
    against tg_conv_dgrad + tg_bn_act_bwd, and both against fp64."""
    from tg_hip import ops as O
    B, H, W, Cc = shape
    g = torch.Generator().manual_seed(sum(shape))
    y = (torch.randn(B, H, W, Cc, generator=g) * 1.5 + 0.3).to(dev)
    gamma, beta = (torch.rand(Cc, generator=g) + 0.5).to(dev), (torch.randn(Cc, generator=g) * 0.3).to(dev)
    ratio = (torch.rand(B, H, W, generator=g) * 2).to(dev)
    w = (torch.randn(1, Cc, 3, 3, generator=g) * 0.2).contiguous(memory_format=torch.channels_last).to(dev)
    dz = torch.randn(B, H, W, 1, generator=g).to(dev)
    mean, rstd = O.bn_stats(y)
    assert O.bn_bwd_conv1_supported(tuple(y.shape))
    assert not O.bn_bwd_conv1_supported((1, 16, 64, 128))          # a small map: the one-launch BatchNorm backward keeps it
    da = O.conv_dgrad(dz, w, tuple(y.shape), 3, 1, 1)
    dy0, dg0, db0, dbias0 = O.bn_act_bwd(da.clone(), y, mean, rstd, gamma, beta, O.ACT_RELU, ratio=ratio, inplace=False)
    dy1, dg1, db1, dbias1 = O.bn_act_bwd_conv1(dz, w, y, mean, rstd, gamma, beta, O.ACT_RELU, ratio=ratio)
    for a, b in ((dy0, dy1), (dg0, dg1), (db0, db1), (dbias0, dbias1)):
        sc = float(a.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * sc + 1e-7, (float((a - b).abs().max()), sc)
    # fp64 reference of the whole chain
    yd, m64, r64 = y.double().cpu(), mean.double().cpu(), rstd.double().cpu()
    dad = F.conv_transpose2d(nchw(dz.double().cpu()), w.double().cpu(), None, 1, 1).permute(0, 2, 3, 1)
    xh = (yd - m64) * r64
    gg = dad * ((xh * gamma.double().cpu() + beta.double().cpu()) > 0)
    n = B * H * W
    dbeta, dgamma = gg.sum((0, 1, 2)), (gg * xh).sum((0, 1, 2))
    dyr = gamma.double().cpu() * r64 * (gg - dbeta / n - xh * dgamma / n) * ratio.double().cpu()[..., None]
    close(dy1, dyr, rtol=1e-4, atol=1e-5 * float(dyr.abs().max()))
    close(dg1, dgamma, rtol=1e-4, atol=1e-5 * float(dgamma.abs().max()))
    close(db1, dbeta, rtol=1e-4, atol=1e-5 * float(dbeta.abs().max()))
    close(dbias1, dyr.sum((0, 1, 2)), rtol=1e-4, atol=2e-5 * float(dyr.abs().sum((0, 1, 2)).max()))
