from .trainer import PTrainer  # noqa: F401
