"""Drop-in `mvp_gan` package for the TERRA-GAN inpainting hot path on MI355X.

The reference's mvp_gan/__init__.py:1 imports its MLflow ExperimentTracker unconditionally, which
drags mlflow/psutil/gitpython into every `import mvp_gan...`.  Tracking is optional here: the name
is re-exported when the reference's `utils.experiment_tracking` is importable, else it is None and
`train()` accepts any duck-typed tracker object."""
try:  # pragma: no cover - only when run inside the reference tree
    from utils.experiment_tracking import ExperimentTracker
except Exception:  # noqa: BLE001
    ExperimentTracker = None

__all__ = ["ExperimentTracker"]
