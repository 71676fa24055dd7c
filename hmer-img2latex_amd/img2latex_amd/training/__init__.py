"""Training step of the hot path (reference img2latex/training/trainer.py:303-343, fp32 branch)."""
from .dp import OverlappedAllReduce, all_reduce_gradients, shard_batch
from .metrics import bleu_n_score, calculate_metrics, levenshtein_distance, masked_accuracy, token_list_accuracy
from .predictor import Predictor, TokenTable, save_checkpoint
from .train_step import TrainStep

__all__ = ["TrainStep", "Predictor", "TokenTable", "save_checkpoint", "all_reduce_gradients", "shard_batch",
           "calculate_metrics", "levenshtein_distance", "bleu_n_score", "masked_accuracy", "token_list_accuracy"]
