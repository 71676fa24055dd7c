"""Same public names as the reference's img2latex.model (model/__init__.py:1-3)."""
from .cnn_encoder import CNNEncoder
from .lstm_decoder import Attention, LSTMDecoder
from .resnet_encoder import ResNetEncoder
from .seq2seq_model import Seq2SeqModel

__all__ = ["CNNEncoder", "ResNetEncoder", "LSTMDecoder", "Attention", "Seq2SeqModel"]
