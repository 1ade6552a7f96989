"""Quality metrics of the training / validation logging path on MI355X.

Mirror of the parts of /root/reference/mvp_gan/src/evaluation/metrics.py that the train loop touches
(`calculate_boundary_quality`, :79-133 -- called from train.py:229-257 and human_guided_trainer.py) and of the tracker's
per-batch figures (utils/experiment_tracking.py:176-231,678-695: PSNR, 11x11 avg-pool SSIM, L1, L2; the same arithmetic as
`MaskEvaluator._calculate_psnr/_calculate_ssim`, metrics.py:47-76).  All of them come from ONE launch pair of the
tg_quality_metrics kernel (tg_hip.ops.quality_metrics): a single read of (pred, target, mask), no intermediate tensors,
and a single device->host copy where the reference issues ~25 ATen ops and 5 `.item()` syncs.

The OpenCV half of the reference file (`MaskEvaluator._identify_features`, IoU/precision/recall of detections) is
post-hoc statistics and out of scope."""
from typing import Dict

import torch

from tg_hip import ops as O

from ..models._common import require_hip


def quality_metrics_tensor(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """9-element DEVICE tensor in tg_hip.ops.QUALITY_KEYS order -- no host synchronisation (accumulate these over a
    validation pass and read them back once)."""
    require_hip(pred, "quality_metrics")
    return O.quality_metrics(pred.detach().contiguous(), target.detach().contiguous(), mask.detach().contiguous().float())


def quality_metrics(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor) -> Dict[str, float]:
    vals = quality_metrics_tensor(pred, target, mask).tolist()            # the one host sync
    return dict(zip(O.QUALITY_KEYS, vals))


def calculate_boundary_quality(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor,
                               boundary_width: int = 10) -> Dict[str, float]:
    """Same name, arguments and keys as the reference (metrics.py:79-133).  `boundary_width` is accepted and unused,
    exactly as there: the band is the 3x3 morphological gradient of the mask."""
    q = quality_metrics(pred, target, mask)
    return {k: q[k] for k in ("boundary_mse", "boundary_psnr", "boundary_gradient_diff")}


def calculate_psnr(pred: torch.Tensor, target: torch.Tensor) -> float:
    """experiment_tracking.py:196-206: 20*log10(1/sqrt(mse)), inf for identical tensors."""
    return quality_metrics(pred, target, torch.ones_like(pred))["psnr"]


def calculate_ssim(pred: torch.Tensor, target: torch.Tensor, window_size: int = 11) -> float:
    """experiment_tracking.py:209-231 (window 11: the only size the reference ever uses)."""
    if window_size != 11:
        raise NotImplementedError("calculate_ssim: the HIP kernel implements the reference's 11x11 window")
    return quality_metrics(pred, target, torch.ones_like(pred))["ssim"]


def calculate_l1_l2(pred: torch.Tensor, target: torch.Tensor):
    """experiment_tracking.py:176-192: (mean |d|, sqrt(mean d^2))."""
    q = quality_metrics(pred, target, torch.ones_like(pred))
    return q["l1_distance"], q["l2_distance"]


def performance_metrics(pred: torch.Tensor, target: torch.Tensor) -> Dict[str, float]:
    """The pred/target part of ExperimentTracker._calculate_performance_metrics (experiment_tracking.py:678-690)."""
    q = quality_metrics(pred, target, torch.ones_like(pred))
    return {"psnr": q["psnr"], "ssim": q["ssim"], "l1_distance": q["l1_distance"], "l2_distance": q["l2_distance"]}
