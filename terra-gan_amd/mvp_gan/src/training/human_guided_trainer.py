"""HumanGuidedTrainer -- generator-only fine-tune with HumanGuidedLoss on MI355X.

Mirror of /root/reference/mvp_gan/src/training/human_guided_trainer.py:29-262: same constructor
(`config, experiment_tracker=None`), `train(generator, train_dataset, num_epochs, checkpoint_dir)`,
Adam(lr = config.training.modes.human_guided.learning_rate), batch size from the same config node,
checkpoint keys `model_state_dict / optimizer_state_dict / loss / config`, and return dictionary.
The step (G forward -> HumanGuidedLoss -> backward -> Adam, :112-153) runs on the HIP engines
(`human_guided_step`).  The reference's blanket per-batch try/except is not reproduced.
"""
import logging
import time
from pathlib import Path

import torch
from torch.utils.data import DataLoader

from tg_hip import engine as E
from tg_hip import ops as O
from tg_hip.gradbuf import grad_buffers

from ..models._common import as_bhw, require_hip
from ..train import _assign_grads, hip_adam_step
from ..utils.losses import HumanGuidedLoss, criterion_forward

logger = logging.getLogger(__name__)


def human_guided_step(generator, criterion, optimizer, images, masks, human_masks=None, grad_sync=None):
    """One fine-tune step (human_guided_trainer.py:112-153).  Returns (loss 1-elem device tensor, generated)."""
    require_hip(images, "human_guided_step")
    GP = generator._tensors()
    img, mask = as_bhw(images, "human_guided_step"), as_bhw(masks, "human_guided_step")
    B, H, W = img.shape
    gen, gctx = E.generator_forward(GP, O.mul(img, mask), mask, generator.training)
    wb, wh = criterion.base_loss_weight, criterion.human_feedback_weight
    total, _parts, dgen = criterion_forward(criterion, gen, img, mask, want_grad=True, scale=wb)
    if human_masks is not None:
        h = (as_bhw(human_masks, "human_guided_step") > 0).float()          # losses.py:168
        out5, _ = O.pixel_losses(gen, img, h, wh, 0.0, wh * max(criterion.boundary_weight, 0.0), l1_weight=h, dpred=dgen,
                                 accumulate=True, eps=criterion.boundary_loss.epsilon)
        total = O.lincomb(total, 1.0, out5[4:5], 1.0)
    optimizer.zero_grad()
    grads, _ = E.generator_backward(GP, gctx, dgen, gbuf=grad_buffers(generator).views)
    _assign_grads(generator, grads)
    buckets = grad_sync(generator, "G") if grad_sync is not None else None
    hip_adam_step(optimizer, grad_sync.grad_scale if grad_sync is not None else 1.0, buckets)
    return total, gen.reshape(B, 1, H, W)


class HumanGuidedTrainer:
    def __init__(self, config, experiment_tracker=None):
        self.config = config
        self.experiment_tracker = experiment_tracker
        if not torch.cuda.is_available():
            raise RuntimeError("HumanGuidedTrainer: no HIP device visible; this build has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device())

    def train(self, generator, train_dataset, num_epochs: int, checkpoint_dir: Path):
        hg = self.config["training"]["modes"]["human_guided"]
        generator = generator.to(self.device)
        criterion = HumanGuidedLoss(self.config, device=self.device)
        optimizer = torch.optim.Adam(generator.parameters(), lr=hg["learning_rate"])
        loader = DataLoader(train_dataset, batch_size=hg["batch_size"], shuffle=True, num_workers=0, pin_memory=False)
        checkpoint_dir = Path(checkpoint_dir)
        best_loss, start_time, epoch = float("inf"), time.time(), 0
        log_interval = self.config["training"].get("log_interval", 10)
        for epoch in range(num_epochs):
            generator.train()
            epoch_start, losses = time.time(), []
            for batch_idx, batch in enumerate(loader):
                images, masks = batch["image"].to(self.device), batch["mask"].to(self.device)
                human = batch.get("human_mask")
                human = human.to(self.device) if human is not None else None
                loss, generated = human_guided_step(generator, criterion, optimizer, images, masks, human)
                losses.append(loss)
                if self.experiment_tracker is not None and batch_idx % log_interval == 0:
                    self.experiment_tracker.log_training_batch(pred=generated, target=images, model=generator,
                                                               optimizer=optimizer, batch_metrics={"loss": float(loss)},
                                                               step=epoch * len(loader) + batch_idx)
            vals = [float(v) for v in losses]                               # one host sync per epoch
            ok = [v for v in vals if v == v and abs(v) != float("inf")]     # NaN/Inf excluded from the mean (:146-148)
            avg = sum(ok) / max(1, len(ok)) if sum(ok) > 0 else 0.0
            epoch_time = time.time() - epoch_start
            if self.experiment_tracker is not None:
                self.experiment_tracker.log_metrics({"epoch.loss": float(avg), "epoch.time": float(epoch_time),
                                                     "epoch.success_rate": float(len(ok) / max(1, len(vals)))}, step=epoch)
            checkpoint = {"epoch": epoch, "model_state_dict": generator.state_dict(),
                          "optimizer_state_dict": optimizer.state_dict(), "loss": float(avg), "config": self.config}
            torch.save(checkpoint, checkpoint_dir / f"generator_epoch_{epoch}.pth")
            if avg < best_loss and avg > 0:
                best_loss = avg
                torch.save(checkpoint, checkpoint_dir / "best_model.pth")
                if self.experiment_tracker is not None and hasattr(self.experiment_tracker, "log_model"):
                    self.experiment_tracker.log_model(generator, "best_human_guided_model", metrics={"loss": float(best_loss)})
            logger.info(f"Epoch {epoch}: loss={avg:.6f}, success_rate={len(ok)}/{len(vals)}, time={epoch_time:.2f}s")
        total_time = time.time() - start_time
        if self.experiment_tracker is not None:
            self.experiment_tracker.log_metrics({"training.total_time": float(total_time), "training.best_loss": float(best_loss)})
        return {"best_loss": best_loss, "total_time": total_time, "final_epoch": epoch, "success": True}
