"""CPU-only checks of the drop-in boundary: the library builds/loads, exports every symbol that
include/terragan_hip.h declares (no compute calls without a GPU), and the Python mirror keeps the
reference's constructor signatures, state-dict keys and seeded initialisation."""
import inspect
import os
import re

import numpy as np
import pytest
import torch

from tests import golden_util as GU

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "terragan_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    import __graft_entry__ as ge
    ge.build()
    from tg_hip import lib as L
    lib = L.load()
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in terragan_hip.h but not exported"
        assert s in L.SIGNATURES, f"{s} has no ctypes signature in tg_hip/lib.py"
    assert sorted(L.SIGNATURES) == syms, set(L.SIGNATURES) ^ set(syms)
    assert lib.tg_version() >= 100


def test_argument_validation_without_gpu():
    """Shape/pointer validation happens on the host before any launch, so it is testable here."""
    import ctypes as C
    from tg_hip import lib as L
    lib = L.load()
    g = L.TgConv(1, 8, 8, 4, 3, 3, 4, 3, 1, 1)              # Ho/Wo inconsistent with H,W,k,s,p
    rc = lib.tg_conv_fwd(C.byref(g), None, None, None, None, None, 0, 0.0, None, None, 0, None)
    assert rc == -1 and b"inconsistent" in lib.tg_last_error()
    rc = lib.tg_bn_stats(None, 10, 8, 1e-5, 0.1, None, None, None, None, None, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.tg_last_error()
    assert lib.tg_conv_wgrad_ws_bytes(C.byref(L.TgConv(2, 16, 16, 64, 16, 16, 64, 3, 1, 1))) > 0


def test_no_cpu_fallback():
    from mvp_gan.src.models import PConvUNet
    from tg_hip import lib as L
    G = PConvUNet()
    with pytest.raises(L.TgError, match="no CPU path"):
        G(torch.zeros(1, 1, 256, 256), torch.ones(1, 1, 256, 256))


def test_constructor_signatures_and_state_dict_contract():
    from mvp_gan.src.models import Discriminator, PConv2d, PConvUNet
    from mvp_gan.src.train import train
    from mvp_gan.src.training.human_guided_trainer import HumanGuidedTrainer
    from mvp_gan.src.utils.losses import BoundaryAwareLoss, HumanGuidedLoss, InpaintingLoss
    assert list(inspect.signature(PConv2d.__init__).parameters)[1:] == \
        ["in_channels", "out_channels", "kernel_size", "stride", "padding", "batch_norm"]
    assert list(inspect.signature(PConvUNet.__init__).parameters) == ["self"]
    assert list(inspect.signature(Discriminator.__init__).parameters) == ["self", "input_channels"]
    assert list(inspect.signature(InpaintingLoss.__init__).parameters)[1:5] == \
        ["perceptual_weight", "tv_weight", "boundary_weight", "device"]
    assert list(inspect.signature(BoundaryAwareLoss.__init__).parameters)[1:] == ["boundary_width", "epsilon", "device"]
    assert list(inspect.signature(HumanGuidedLoss.forward).parameters) == ["self", "input", "target", "mask", "human_feedback"]
    assert list(inspect.signature(train).parameters)[:11] == \
        ["img_dir", "mask_dir", "generator", "discriminator", "optimizer_G", "optimizer_D", "checkpoint_path", "config",
         "experiment_tracker", "val_img_dir", "val_mask_dir"]
    assert list(inspect.signature(HumanGuidedTrainer.train).parameters) == \
        ["self", "generator", "train_dataset", "num_epochs", "checkpoint_dir"]


def test_seeded_init_matches_reference():
    """torch.manual_seed(0); G, D, criterion in that order -> bit-identical parameters."""
    from mvp_gan.src.models import Discriminator, PConvUNet
    from mvp_gan.src.utils.losses import InpaintingLoss
    gold = GU.load("init")
    torch.manual_seed(0)
    G, D = PConvUNet(), Discriminator()
    crit = InpaintingLoss(0.1, 0.1, device=torch.device("cpu"))
    for pre, mod in [("G", G), ("D", D), ("V", crit.vgg_layers)]:
        sd = mod.state_dict()
        assert list(sd.keys()) == [str(k) for k in gold[f"{pre}/keys"]], pre
        assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in gold[f"{pre}/shapes"]]
        for k, v in sd.items():
            if v.dtype.is_floating_point:
                assert np.array_equal(v.flatten()[:8].numpy(), gold[f"{pre}/first/{k}"]), k     # logical OIHW order
                ref = gold[f"{pre}/w/{k}"]
                assert float(v.double().sum()) == ref[0], k
    # conv weights are stored [Cout][kh][kw][Cin] without changing their logical shape
    w = G.enc2.input_conv.weight
    assert tuple(w.shape) == (128, 64, 5, 5) and w.permute(0, 2, 3, 1).is_contiguous()
    # a reference-style (contiguous OIHW) checkpoint loads into the channels_last storage
    sd = {k: v.clone().contiguous() for k, v in G.state_dict().items()}
    G2 = PConvUNet()
    G2.load_state_dict(sd)
    assert torch.equal(G2.enc2.input_conv.weight, w) and G2.enc2.input_conv.weight.permute(0, 2, 3, 1).is_contiguous()


def test_bench_input_recipe_matches_oracle_copy():
    """bench.py draws its synthetic tiles from tg_hip.synth (the product must not import oracle/); the oracle keeps its
    own copy of the SURVEY §8d recipe for the fixtures.  They must agree bit for bit."""
    from oracle import terragan_oracle as Orc
    from tg_hip.synth import synth_batch
    for b, size, seed in [(2, 64, 3), (1, 256, 1000)]:
        a, m = synth_batch(b, size, seed)
        a2, m2 = Orc.synth_batch(b, size, seed)
        assert torch.equal(a, a2) and torch.equal(m, m2)


def test_vgg_trunk_weights_resolution(tmp_path, monkeypatch):
    """losses.py:31-34: the reference loads torchvision's ImageNet VGG16.  Offline that cannot work: the mirror must (i) RAISE
    unless the caller opts in to the stand-in trunk, (ii) load a local vgg16-format state-dict through
    TERRAGAN_VGG16_WEIGHTS / vgg_weights=, keeping only features[:16] (keys `features.N.*`, N < 16)."""
    import pytest
    from mvp_gan.src.utils.losses import InpaintingLoss
    monkeypatch.delenv("TERRAGAN_VGG16_WEIGHTS", raising=False)
    monkeypatch.setenv("TERRAGAN_ALLOW_STANDIN_VGG", "0")
    with pytest.raises(RuntimeError, match="TERRAGAN_VGG16_WEIGHTS"):
        InpaintingLoss(0.1, 0.1, device=torch.device("cpu"))
    InpaintingLoss(0.1, 0.1, device=torch.device("cpu"), allow_standin_vgg=True)           # explicit opt-in
    # a synthetic torchvision-format vgg16 state-dict: all 13 convs + classifier keys
    g = torch.Generator().manual_seed(3)
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    sd, idx, cin = {}, 0, 3
    for v in cfg:
        if v == "M":
            idx += 1
            continue
        sd[f"features.{idx}.weight"] = torch.randn(v, cin, 3, 3, generator=g) * 0.01
        sd[f"features.{idx}.bias"] = torch.randn(v, generator=g) * 0.01
        idx, cin = idx + 2, v
    sd["classifier.0.weight"], sd["classifier.0.bias"] = torch.zeros(4, 4), torch.zeros(4)
    path = tmp_path / "vgg16.pth"
    torch.save(sd, path)
    monkeypatch.setenv("TERRAGAN_VGG16_WEIGHTS", str(path))
    crit = InpaintingLoss(0.1, 0.1, device=torch.device("cpu"))
    got = crit.vgg_layers.state_dict()
    assert sorted(got) == sorted(f"{i}.{p}" for i in (0, 2, 5, 7, 10, 12, 14) for p in ("weight", "bias"))
    for k, v in got.items():
        assert torch.equal(v, sd["features." + k]), k
    assert not any(p_.requires_grad for p_ in crit.vgg_layers.parameters())
    crit2 = InpaintingLoss(0.1, 0.1, device=torch.device("cpu"), vgg_weights=str(path))   # explicit argument, same result
    assert torch.equal(crit2.vgg_layers.state_dict()["14.weight"], sd["features.14.weight"])
