from .human_guided_trainer import HumanGuidedTrainer

__all__ = ["HumanGuidedTrainer"]
