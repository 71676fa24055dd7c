"""img2latex_amd -- MI355X-native hot path of hmer-img2latex (encoder + LSTM decoder + search)."""
__version__ = "0.1.0"
