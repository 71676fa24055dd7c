"""Training step of the hot path (reference img2latex/training/trainer.py:303-343, fp32 branch)."""
from .dp import all_reduce_gradients, shard_batch
from .train_step import TrainStep

__all__ = ["TrainStep", "all_reduce_gradients", "shard_batch"]
