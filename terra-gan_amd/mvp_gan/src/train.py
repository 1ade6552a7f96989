"""GAN training loop of the inpainting hot path on MI355X.

`train(...)` keeps the signature, config handling (.get() defaults: batch 2, lr 2e-4, 10 epochs,
log_interval 10, checkpoint_interval 5), checkpoint dictionary and return value of
/root/reference/mvp_gan/src/train.py:23-453.  The loop body (train.py:177-225) is factored out as
`train_step(...)`, which drives the HIP engines directly in the reference's order:

    optG.zero_grad -> G(masked, mask) -> criterion -> D(gen) -> BCE(.,1) -> backward -> optG.step
    optD.zero_grad -> D(real), D(gen.detach()) -> 0.5*(BCE(.,1)+BCE(.,0)) -> backward -> optD.step

Differences from the reference, none of which changes a result:
  * loss scalars stay on the device; nothing in the step synchronises with the host (the reference
    syncs 7x per step: train.py:222-225, losses.py:111,411,419);
  * the discriminator parameter gradients that the reference computes in the generator step and
    then discards (train.py:204 -> 210) are not computed;
  * D(gen.detach()) in the discriminator step has exactly the inputs and weights of D(gen) in the
    generator step (D is updated only afterwards), so its activations are reused and only its
    BatchNorm running-stat side effect is replayed (`reuse_fake_forward=True`);
  * torch.optim.Adam's arithmetic runs in the tg_adam HIP kernel on the optimiser's own state
    tensors (state-dict compatible);
  * no blanket try/except around the batch (train.py:178,268-270): errors raise.
Data parallelism: pass `grad_sync` (see tg_hip.dist.GradSync) to all-reduce G and D gradients
over RCCL before each Adam step.  `train(..., grad_sync=...)` then also (i) broadcasts rank 0's parameters,
buffers and optimiser state before the first step, (ii) gives every rank its own shard of the dataset
(tg_hip.dist.ShardSampler, one shared permutation per epoch), (iii) writes checkpoints and tracker
records on rank 0 only.  BatchNorm statistics stay per rank (no SyncBN in the reference); rank 0's are saved.
"""
import logging
import time
from pathlib import Path
from typing import Dict, Optional

import torch
from torch.utils.data import DataLoader

from tg_hip import engine as E
from tg_hip import ops as O
from tg_hip.gradbuf import grad_buffers

from mvp_gan import ExperimentTracker  # noqa: F401  (None when tracking is unavailable)
from .models._common import as_bhw, require_hip
from .models.discriminator import Discriminator
from .models.generator import PConvUNet
from .utils.dataset import InpaintingDataset, resize_to_tensor
from .utils.shard_dataset import ShardLoader, is_shard
from .evaluation.metrics import calculate_boundary_quality
from .utils.losses import HumanGuidedLoss, InpaintingLoss, criterion_forward  # noqa: F401

logger = logging.getLogger(__name__)


# --------------------------------------------------------------------------------------------------
# Adam on the HIP kernel, operating on a torch.optim.Adam's own parameter groups and state
# --------------------------------------------------------------------------------------------------
def hip_adam_step(optimizer, grad_scale=1.0, buckets=None):
    """optimizer.step() for torch.optim.Adam (defaults) with the update arithmetic in the tg_adam kernel,
    on the optimiser's own state tensors.  `buckets` (from tg_hip.dist.GradSync) orders the walk and
    inserts the per-bucket wait so that later buckets' all-reduce overlaps earlier buckets' update."""
    if not isinstance(optimizer, torch.optim.Adam) or isinstance(optimizer, torch.optim.AdamW):
        logger.warning("hip_adam_step: %s is not torch.optim.Adam; delegating to its own step()", type(optimizer).__name__)
        for b in buckets or []:
            b.wait()
        optimizer.step()
        return
    hyper = {}
    for group in optimizer.param_groups:
        if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
            raise NotImplementedError("hip_adam_step supports torch.optim.Adam defaults (no weight decay / amsgrad / maximize)")
        for p in group["params"]:
            hyper[p] = (float(group["lr"]), group["betas"][0], group["betas"][1], float(group["eps"]))

    layout_ok = optimizer.__dict__.setdefault("_tg_layout_ok", set())     # (p, m, v) triples already checked

    def _update(plist):
        """One multi-tensor launch per (hyper-parameter set, step count) group -- normally a single group."""
        groups = {}
        for p in plist:
            if p.grad is None or p not in hyper:
                continue
            st = optimizer.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif (p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) not in layout_ok:
                # a state loaded from a reference-format checkpoint (main_pipeline.py:260-263) keeps the SAVED strides --
                # contiguous OIHW -- while the parameters here are stored channels_last: re-lay the moments once to the
                # parameter's physical layout (the kernel walks plain memory)
                for mk in ("exp_avg", "exp_avg_sq"):
                    t = st[mk]
                    if t.shape != p.shape:
                        raise ValueError(f"hip_adam_step: optimizer state {mk} has shape {tuple(t.shape)}, parameter {tuple(p.shape)}")
                    if t.device != p.device or t.dtype != p.dtype or not (O._dense_layouts(t) & O._dense_layouts(p)):
                        new = torch.empty_like(p, memory_format=torch.preserve_format)
                        new.copy_(t)
                        st[mk] = new
                if not torch.is_tensor(st["step"]):
                    st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
                layout_ok.add((p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()))
            st["step"] += 1
            groups.setdefault(hyper[p] + (int(st["step"]),), []).append((p, st))
        for (lr, b1, b2, eps, step), items in groups.items():
            O.adam_multi_([p.data for p, _ in items], [p.grad for p, _ in items], [st["exp_avg"] for _, st in items],
                          [st["exp_avg_sq"] for _, st in items], lr, b1, b2, eps, step, grad_scale)

    if buckets is None:
        _update(list(hyper))
    else:
        for b in buckets:
            b.wait()
            _update(b.params)


def _assign_grads(module, grads):
    for k, p in module.named_parameters():
        g = grads.get(k)
        if g is not None:
            p.grad = g


def train_step(generator, discriminator, criterion, optimizer_G, optimizer_D, real_imgs, masks, grad_sync=None,
               reuse_fake_forward=True):
    """One GAN step (reference loop body, train.py:177-219).  real_imgs/masks: [B,1,H,W] on the GPU.
    Returns a dict of 1-element device tensors (g_total, g_loss, g_adv, d_loss, real_loss, fake_loss)
    plus 'gen' ([B,1,H,W], detached)."""
    try:
        return _train_step(generator, discriminator, criterion, optimizer_G, optimizer_D, real_imgs, masks, grad_sync,
                           reuse_fake_forward)
    except BaseException:
        if grad_sync is not None:          # collectives launched by a step that raised are never waited for
            grad_sync.reset()
        raise


def _train_step(generator, discriminator, criterion, optimizer_G, optimizer_D, real_imgs, masks, grad_sync, reuse_fake_forward):
    require_hip(real_imgs, "train_step")
    GP, DP = generator._tensors(), discriminator._tensors()
    real, mask = as_bhw(real_imgs, "train_step"), as_bhw(masks, "train_step")
    B, H, W = real.shape
    # [gen; real] stacked along the batch: what the perceptual trunk (losses.py:79-88) and the grouped discriminator forward
    # read.  real lands there from the pass that computes real*mask, gen is written there by the generator itself.
    both = torch.empty((2 * B, H, W), dtype=torch.float32, device=real.device)
    masked = O.mul(real, mask, keep=both[B:])                                # train.py:181

    # ---- generator ------------------------------------------------------------------------------
    optimizer_G.zero_grad()                                                  # set_to_none (App. A #12)
    gen, gctx = E.generator_forward(GP, masked, mask, generator.training,
                                    checkpoint=getattr(generator, "activation_checkpointing", False), out=both[:B])     # train.py:185
    if grad_sync is not None and grad_sync._inflight != 0:
        # The loss stack's dgrads run on wino44_kernel, the one persistent Winograd kernel WITHOUT a work-stealing variant
        # (csrc/wino44.inc): that is sound only while no gradient bucket is in flight here -- every bucket of the previous step
        # was waited for by its hip_adam_step, and this step's first bucket is launched inside generator_backward below.
        raise RuntimeError(f"train_step: {grad_sync._inflight} gradient collectives still in flight at the loss stack")
    g_loss, _parts, dgen = criterion_forward(criterion, gen, real, mask, want_grad=True, both=both)   # train.py:188
    # D(fake) of this step (train.py:202), D(real) and D(fake.detach()) of the discriminator step (train.py:211-212) see the
    # same discriminator weights, and D(real) depends on nothing the generator step produces: the two distinct passes are
    # stacked along the batch and run as ONE grouped forward (convolutions once over 2B images, BatchNorm statistics per
    # pass), and the discriminator step's two backward passes as one.  Running statistics are replayed in the
    # reference's order fake, real, fake.
    grouped = reuse_fake_forward and discriminator.training
    if grouped:
        logits2, dctx2 = E.discriminator_forward(DP, both, True, groups=2, update_running=False)
        E.discriminator_replay_running_stats(DP, dctx2, order=(0,))          # train.py:202
        fake_logits, real_logits = logits2[:B], logits2[B:]
        dctx_fake = E.discriminator_group(dctx2, 0)
    else:
        fake_logits, dctx_fake = E.discriminator_forward(DP, gen, discriminator.training)  # train.py:202
    g_adv, dlogits = O.bce_logits(fake_logits, 1.0)                          # train.py:203
    g_total = O.lincomb(g_loss, 1.0, g_adv, 1.0)                             # train.py:204
    _none, dgen_adv = E.discriminator_backward(DP, dctx_fake, dlogits, want_wgrad=False, want_dimg=True)
    O.axpby_(dgen_adv.reshape(B, H, W), 1.0, 1.0, dgen)
    # Data parallel: every ~25 MB bucket of the generator's gradients is all-reduced as soon as its last weight gradient
    # has been enqueued (BucketLauncher.ready, called by the backward schedule), so the exchange runs on RCCL's stream
    # underneath the rest of the backward and the whole discriminator step below, which needs neither the reduced
    # gradients nor the updated generator weights (it consumes gen.detach() and D's own parameters);
    # optimizer_G.step() is applied after it.  Same results.
    gl = grad_sync.begin(generator, "G") if grad_sync is not None else None
    ggrads, _ = E.generator_backward(GP, gctx, dgen, gbuf=grad_buffers(generator).views,
                                     on_ready=gl.ready if gl is not None else None)        # train.py:206
    del gctx
    _assign_grads(generator, ggrads)
    gb = gl.finish() if gl is not None else None
    if grad_sync is None:
        hip_adam_step(optimizer_G)                                           # train.py:207

    # ---- discriminator ----------------------------------------------------------------------------
    optimizer_D.zero_grad()                                                  # train.py:210
    if grouped:
        E.discriminator_replay_running_stats(DP, dctx2, order=(1, 0))        # train.py:211, 212
    else:
        real_logits, dctx_real = E.discriminator_forward(DP, real, discriminator.training)     # train.py:211
        fake_logits, dctx_fake = E.discriminator_forward(DP, gen, discriminator.training)      # train.py:212
    dl2 = torch.empty_like(logits2) if grouped else None                     # [dl_fake; dl_real], the order of the grouped forward
    real_loss, dl_real = O.bce_logits(real_logits, 1.0, coef=0.5, dz_out=dl2[B:] if grouped else None)     # train.py:215,217
    fake_loss, dl_fake = O.bce_logits(fake_logits, 0.0, coef=0.5, dz_out=dl2[:B] if grouped else None)     # train.py:216,217
    d_loss = O.lincomb(real_loss, 0.5, fake_loss, 0.5)
    dl_ = grad_sync.begin(discriminator, "D") if grad_sync is not None else None
    if grouped:
        dg_real, _ = E.discriminator_backward(DP, dctx2, dl2, want_wgrad=True,
                                              gbuf=grad_buffers(discriminator).views,
                                              on_ready=dl_.ready if dl_ is not None else None)      # train.py:218
    else:
        dg_real, _ = E.discriminator_backward(DP, dctx_real, dl_real, want_wgrad=True,
                                              gbuf=grad_buffers(discriminator).views)      # train.py:218
        dg_fake, _ = E.discriminator_backward(DP, dctx_fake, dl_fake, want_wgrad=True)
        for k, g in dg_real.items():
            O.axpby_(dg_fake[k], 1.0, 1.0, g)
    _assign_grads(discriminator, dg_real)
    if grad_sync is not None:
        db = dl_.finish()
        hip_adam_step(optimizer_G, grad_sync.grad_scale, gb)                 # train.py:207 (deferred, see above)
        hip_adam_step(optimizer_D, grad_sync.grad_scale, db)                 # train.py:219
    else:
        hip_adam_step(optimizer_D)                                           # train.py:219

    return {"gen": gen.reshape(B, 1, H, W), "g_total": g_total, "g_loss": g_loss, "g_adv": g_adv, "d_loss": d_loss,
            "real_loss": real_loss, "fake_loss": fake_loss}


@torch.no_grad()
def validation_step(generator, discriminator, criterion, real_imgs, masks):
    """Validation body (train.py:283-301); note D stays in whatever mode the caller left it in --
    the reference leaves it in train mode, so its BN running stats move here too (App. A #8)."""
    real, mask = as_bhw(real_imgs, "validation_step"), as_bhw(masks, "validation_step")
    GP, DP = generator._tensors(), discriminator._tensors()
    gen, _ = E.generator_forward(GP, O.mul(real, mask), mask, generator.training)
    g_total, _parts, _ = criterion_forward(criterion, gen, real, mask, want_grad=False)
    rl, _ = E.discriminator_forward(DP, real, discriminator.training)
    fl, _ = E.discriminator_forward(DP, gen, discriminator.training)
    d_real, _ = O.bce_logits(rl, 1.0, want_grad=False)
    d_fake, _ = O.bce_logits(fl, 0.0, want_grad=False)
    return g_total, O.lincomb(d_real, 0.5, d_fake, 0.5)


def _default_config():
    return {"training": {"batch_size": 2, "learning_rate": 2e-4, "epochs": 10,
                         "loss_weights": {"perceptual": 0.1, "tv": 0.1}}}


def train(img_dir: Path, mask_dir: Path, generator: Optional[PConvUNet] = None,
          discriminator: Optional[Discriminator] = None, optimizer_G: Optional[torch.optim.Optimizer] = None,
          optimizer_D: Optional[torch.optim.Optimizer] = None, checkpoint_path: Optional[Path] = None,
          config: Optional[Dict] = None, experiment_tracker=None, val_img_dir: Optional[Path] = None,
          val_mask_dir: Optional[Path] = None, grad_sync=None, img_size=(512, 512)):
    if not torch.cuda.is_available():
        raise RuntimeError("mvp_gan.src.train.train: no HIP device visible; this build has no CPU path")
    device = torch.device("cuda", torch.cuda.current_device())
    if config is None:
        config = _default_config()
    tcfg = config["training"]
    transform = resize_to_tensor(img_size)                                   # train.py:67-70
    from tg_hip.dist import ShardSampler, broadcast_state, rank_world
    rank, world = rank_world(getattr(grad_sync, "group", None)) if grad_sync is not None else (0, 1)
    bs = tcfg.get("batch_size", 2)
    if is_shard(str(img_dir)):
        # pre-decoded uint8 shard (utils/shard_dataset.py): `img_dir` is the shard directory, `mask_dir` is ignored;
        # /255 and the mask binarisation run on the device, bit-identical to the PNG path below
        train_loader = ShardLoader(str(img_dir), bs, shuffle=True, device=device)
        sampler = ShardSampler(train_loader.n, rank, world, shuffle=True, seed=int(tcfg.get("seed", 0))) if world > 1 else None
        train_loader.sampler = sampler
    else:
        train_set = InpaintingDataset(img_dir, mask_dir, transform=transform)
        sampler = ShardSampler(len(train_set), rank, world, shuffle=True, seed=int(tcfg.get("seed", 0))) if world > 1 else None
        train_loader = DataLoader(train_set, batch_size=bs, shuffle=sampler is None, sampler=sampler, num_workers=0)
    val_loader = None
    if val_img_dir is not None and is_shard(str(val_img_dir)):
        val_loader = ShardLoader(str(val_img_dir), bs, shuffle=False, device=device)
    elif val_img_dir is not None and val_mask_dir is not None:
        val_loader = DataLoader(InpaintingDataset(val_img_dir, val_mask_dir, transform=transform),
                                batch_size=bs, shuffle=False, num_workers=0)
    if generator is None:
        generator = PConvUNet().to(device)
    if discriminator is None:
        discriminator = Discriminator().to(device)
    criterion = InpaintingLoss(perceptual_weight=tcfg["loss_weights"]["perceptual"], tv_weight=tcfg["loss_weights"]["tv"],
                               device=device)                               # boundary weight stays 0.5 (App. A #5)
    if optimizer_G is None:
        optimizer_G = torch.optim.Adam(generator.parameters(), lr=tcfg.get("learning_rate", 2e-4))
    if optimizer_D is None:
        optimizer_D = torch.optim.Adam(discriminator.parameters(), lr=tcfg.get("learning_rate", 2e-4))
    if world > 1:
        broadcast_state((generator, discriminator), (optimizer_G, optimizer_D), group=getattr(grad_sync, "group", None))
        if rank != 0:                    # checkpoints and tracker records come from rank 0 only
            checkpoint_path, experiment_tracker = None, None
    if experiment_tracker is not None and hasattr(experiment_tracker, "_log_model_architecture"):
        experiment_tracker._log_model_architecture(generator)

    log_interval = tcfg.get("log_interval", 10)
    boundary_cfg = tcfg["loss_weights"].get("boundary", 0.0)
    best_val_loss, best_train_loss = float("inf"), float("inf")
    start_time, epoch, checkpoint, val_g_loss = time.time(), -1, None, float("inf")
    for epoch in range(tcfg.get("epochs", 10)):
        generator.train()
        discriminator.train()
        keys = ["g_loss", "d_loss", "real_loss", "fake_loss"]
        sums = {k: torch.zeros(1, device=device) for k in keys}          # device-side running sums
        epoch_start = time.time()
        if sampler is not None:
            sampler.set_epoch(epoch)
        for batch_idx, data in enumerate(train_loader):
            real = data["image"].to(device, non_blocking=True)
            masks = data["mask"].to(device, non_blocking=True)
            out = train_step(generator, discriminator, criterion, optimizer_G, optimizer_D, real, masks, grad_sync)
            for k, src in zip(keys, ["g_total", "d_loss", "real_loss", "fake_loss"]):
                O.axpby_(out[src], 1.0, 1.0, sums[k])
            if experiment_tracker is not None and batch_idx % log_interval == 0:
                bm = {k: float(out[s_]) for k, s_ in zip(keys, ["g_total", "d_loss", "real_loss", "fake_loss"])}
                if boundary_cfg > 0:                     # train.py:235-257: boundary quality + the boundary loss term
                    bm.update(calculate_boundary_quality(out["gen"], real, masks))
                    bl = float(criterion.boundary_loss(out["gen"], real, masks))
                    if bl > 0:
                        bm["boundary_loss"] = bl
                experiment_tracker.log_training_batch(pred=out["gen"], target=real, model=generator, optimizer=optimizer_G,
                                                      batch_metrics=bm, step=epoch * len(train_loader) + batch_idx)
        nb = max(len(train_loader), 1)
        epoch_metrics = {k: float(v) / nb for k, v in sums.items()}      # one host sync per epoch
        epoch_metrics["epoch_time"] = time.time() - epoch_start

        if val_loader is not None:
            generator.eval()
            vg, vd = torch.zeros(1, device=device), torch.zeros(1, device=device)
            for vb in val_loader:
                g_, d_ = validation_step(generator, discriminator, criterion, vb["image"].to(device), vb["mask"].to(device))
                O.axpby_(g_, 1.0, 1.0, vg)
                O.axpby_(d_, 1.0, 1.0, vd)
            val_g_loss, val_d_loss = float(vg) / len(val_loader), float(vd) / len(val_loader)
            if experiment_tracker is not None:
                experiment_tracker.log_metrics({"validation.g_loss": val_g_loss, "validation.d_loss": val_d_loss}, step=epoch)
            improved = val_g_loss < best_val_loss
            if improved:
                best_val_loss = val_g_loss
        else:
            improved = epoch_metrics["g_loss"] < best_train_loss
            if improved:
                best_train_loss = epoch_metrics["g_loss"]
        if improved:
            checkpoint = {"epoch": epoch, "generator_state_dict": generator.state_dict(),
                          "discriminator_state_dict": discriminator.state_dict(),
                          "optimizer_G_state_dict": optimizer_G.state_dict(),
                          "optimizer_D_state_dict": optimizer_D.state_dict(),
                          "g_loss": float(epoch_metrics["g_loss"]), "d_loss": float(epoch_metrics["d_loss"]), "config": config}
            if val_loader is not None:
                checkpoint.update(val_g_loss=float(val_g_loss), val_d_loss=float(val_d_loss))
            if checkpoint_path is not None:
                torch.save(checkpoint, checkpoint_path)
            if experiment_tracker is not None and hasattr(experiment_tracker, "log_model"):
                metrics = {k: v for k, v in checkpoint.items() if isinstance(v, (int, float))}
                experiment_tracker.log_model(generator, "best_model_validation" if val_loader is not None else "best_model_train",
                                             metrics=metrics)
        if checkpoint_path is not None and checkpoint is not None and epoch % tcfg.get("checkpoint_interval", 5) == 0:
            torch.save(checkpoint, Path(checkpoint_path).parent / f"checkpoint_epoch_{epoch}.pth")
        if experiment_tracker is not None:
            experiment_tracker.log_metrics({"epoch.g_loss": epoch_metrics["g_loss"], "epoch.d_loss": epoch_metrics["d_loss"],
                                            "epoch.real_loss": epoch_metrics["real_loss"],
                                            "epoch.fake_loss": epoch_metrics["fake_loss"],
                                            "epoch.time": epoch_metrics["epoch_time"]}, step=epoch)
        msg = f"Epoch {epoch}: g_loss={epoch_metrics['g_loss']:.4f}, d_loss={epoch_metrics['d_loss']:.4f}"
        if val_loader is not None:
            msg += f", val_g_loss={val_g_loss:.4f}, val_d_loss={val_d_loss:.4f}"
        logger.info(msg + f", time={epoch_metrics['epoch_time']:.2f}s")

    total_time = time.time() - start_time
    if experiment_tracker is not None:
        fm = {"training.total_time": float(total_time), "training.best_train_loss": float(best_train_loss)}
        if val_loader is not None:
            fm["training.best_val_loss"] = float(best_val_loss)
            fm["training.validation_improvement"] = float(best_val_loss - val_g_loss)
        experiment_tracker.log_metrics(fm)
    return {"best_train_loss": best_train_loss, "best_val_loss": best_val_loss if val_loader is not None else None,
            "total_time": total_time, "final_epoch": epoch}
