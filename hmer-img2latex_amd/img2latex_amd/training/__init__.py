"""Training step of the hot path (reference img2latex/training/trainer.py:303-343, fp32 branch)."""
from .dp import all_reduce_gradients, shard_batch
from .predictor import Predictor, TokenTable, save_checkpoint
from .train_step import TrainStep

__all__ = ["TrainStep", "Predictor", "TokenTable", "save_checkpoint", "all_reduce_gradients", "shard_batch"]
