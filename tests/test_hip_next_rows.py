"""GPU parity of the SURVEY §8f "next" rows that already run on the HIP kernels: batched inference / evaluate(),
the human-guided fine-tune step, and train() end to end on a tiny PNG data set (checkpoint format included)."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from tg_hip import lib
    lib.load()
    return torch.device("cuda:0")


def _oracle_params(mod):
    return {k: v.detach().cpu().clone().contiguous() for k, v in mod.state_dict().items()}


def test_inference_eval_mode_and_evaluate(dev, tmp_path):
    from mvp_gan.src.evaluate import evaluate, inpaint_batch
    from mvp_gan.src.models import PConvUNet
    from oracle import terragan_oracle as Orc
    torch.manual_seed(21)
    G = PConvUNet()
    g = torch.Generator().manual_seed(3)
    for m in G.modules():                       # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    gp = _oracle_params(G)
    G = G.to(dev).eval()
    real, mask = Orc.synth_batch(3, 128, 9)
    out = inpaint_batch(G, real.to(dev), mask.to(dev))
    ref = Orc.generator_forward(gp, real * mask, mask, training=False)
    assert (out.cpu() - ref).abs().max().item() <= 2e-5
    assert int(G.enc1.bn.num_batches_tracked) == 0          # eval mode leaves the statistics alone
    # evaluate(): PNG in -> 500x500 PNG out, uint8 truncation of out*255 then PIL bilinear (evaluate.py:53-59)
    img = (torch.rand(1, 200, 200, generator=g) * 255).byte().numpy()[0]
    msk = np.full((200, 200), 255, np.uint8)
    msk[60:120, 40:150] = 0
    Image.fromarray(img, mode="L").save(tmp_path / "a.png")
    Image.fromarray(msk, mode="L").save(tmp_path / "a_mask.png")
    evaluate(str(tmp_path / "a.png"), str(tmp_path / "a_mask.png"), G, str(tmp_path / "o.png"))
    got = np.asarray(Image.open(tmp_path / "o.png"))
    assert got.shape == (500, 500) and got.dtype == np.uint8
    x = torch.from_numpy(np.asarray(Image.fromarray(img, mode="L").resize((512, 512), Image.BILINEAR), dtype=np.float32) / 255.0)
    m2 = torch.from_numpy((np.asarray(Image.fromarray(msk, mode="L").resize((512, 512), Image.BILINEAR)) > 0).astype(np.float32))
    x, m2 = x[None, None], m2[None, None]
    o = Orc.generator_forward(gp, x * m2, m2, training=False)[0, 0].numpy()
    want = np.asarray(Image.fromarray((o * 255).astype("uint8"), mode="L").resize((500, 500), Image.BILINEAR))
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1      # +-1 LSB where out*255 sits on an integer boundary


def test_human_guided_step(dev):
    from mvp_gan.src.models import PConvUNet
    from mvp_gan.src.training.human_guided_trainer import human_guided_step
    from mvp_gan.src.utils.losses import HumanGuidedLoss
    from oracle import terragan_oracle as Orc
    cfg = {"training": {"loss_weights": {"boundary": 0.5},
                        "modes": {"human_guided": {"human_feedback_weight": 0.3, "base_loss_weight": 0.7,
                                                   "learning_rate": 1e-4, "batch_size": 5}}}}
    torch.manual_seed(0)
    G = PConvUNet()
    crit = HumanGuidedLoss(cfg, device=torch.device("cpu"))
    gp = _oracle_params(G)
    vp = {k: v.detach().clone() for k, v in crit.vgg_layers.state_dict().items()}
    G, crit = G.to(dev), crit.to(dev)
    opt = torch.optim.Adam(G.parameters(), lr=1e-4)
    real, mask = Orc.synth_batch(4, 128, 31)
    _, human = Orc.synth_batch(4, 128, 32)
    human = (1 - human) * 255.0                                   # annotated regions, 8-bit style
    loss, gen = human_guided_step(G, crit, opt, real.to(dev), mask.to(dev), human.to(dev))
    keys = Orc.trainable(gp)
    for k in keys:
        gp[k].requires_grad_(True)
    geno = Orc.generator_forward(gp, real * mask, mask, True)
    lo = Orc.human_guided_loss(vp, geno, real, mask, human)
    grads = torch.autograd.grad(lo, [gp[k] for k in keys])
    assert abs(float(loss) - float(lo)) <= 2e-5 * abs(float(lo))
    assert (gen.cpu() - geno.detach()).abs().max().item() <= 2e-5
    mine = dict(G.named_parameters())
    for k, gr in zip(keys, grads):
        if k.endswith("input_conv.bias"):
            continue
        e = (mine[k].grad.cpu() - gr).abs().max().item()
        assert e <= 2e-2 * gr.abs().max().item() + 1e-7, (k, e, gr.abs().max().item())
    st = opt.state[G.final.weight]
    assert int(st["step"]) == 1


def test_train_end_to_end_tiny_dataset(dev, tmp_path):
    from mvp_gan.src.models import Discriminator, PConvUNet
    from mvp_gan.src.train import train
    rng = np.random.default_rng(1)
    for split in ("tr", "va"):
        (tmp_path / split / "img").mkdir(parents=True)
        (tmp_path / split / "msk").mkdir(parents=True)
        for i in range(4):
            Image.fromarray(rng.integers(0, 256, (100, 100), dtype=np.uint8), mode="L").save(tmp_path / split / "img" / f"t{i}.png")
            m = np.full((100, 100), 255, np.uint8)
            m[20 + 5 * i:60, 30:70] = 0
            Image.fromarray(m, mode="L").save(tmp_path / split / "msk" / f"t{i}.png")

    class Tracker:                       # duck-typed experiment tracker (train.py:151,259,312,...)
        def __init__(self):
            self.batches, self.metrics, self.models = 0, [], 0

        def _log_model_architecture(self, model):
            pass

        def log_training_batch(self, **kw):
            assert set(kw) == {"pred", "target", "model", "optimizer", "batch_metrics", "step"}
            self.batches += 1

        def log_metrics(self, metrics, step=None):
            self.metrics.append(metrics)

        def log_model(self, model, name, metrics=None, **kw):
            self.models += 1

    cfg = {"training": {"batch_size": 2, "learning_rate": 2e-4, "epochs": 2, "log_interval": 1, "checkpoint_interval": 1,
                        "loss_weights": {"perceptual": 0.1, "tv": 0.1, "boundary": 0.5}}}
    ckpt = tmp_path / "best.pth"
    tr = Tracker()
    torch.manual_seed(0)
    res = train(tmp_path / "tr" / "img", tmp_path / "tr" / "msk", checkpoint_path=ckpt, config=cfg, experiment_tracker=tr,
                val_img_dir=tmp_path / "va" / "img", val_mask_dir=tmp_path / "va" / "msk", img_size=(128, 128))
    assert set(res) == {"best_train_loss", "best_val_loss", "total_time", "final_epoch"} and res["final_epoch"] == 1
    assert np.isfinite(res["best_val_loss"]) and tr.batches == 4 and tr.models >= 1
    ck = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert {"epoch", "generator_state_dict", "discriminator_state_dict", "optimizer_G_state_dict", "optimizer_D_state_dict",
            "g_loss", "d_loss", "val_g_loss", "val_d_loss", "config"} <= set(ck)
    G2, D2 = PConvUNet(), Discriminator()
    G2.load_state_dict(ck["generator_state_dict"])
    D2.load_state_dict(ck["discriminator_state_dict"])
    oG = torch.optim.Adam(G2.parameters(), lr=2e-4)
    oG.load_state_dict(ck["optimizer_G_state_dict"])            # the HIP Adam keeps torch's optimizer state format
    assert (tmp_path / "checkpoint_epoch_0.pth").exists()
