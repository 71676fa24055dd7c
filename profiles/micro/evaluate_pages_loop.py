"""40 evaluate batches from ragged pages through Predictor.evaluate_stream (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
import numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "hmer-img2latex_amd"))
from img2latex_amd import synth
from img2latex_amd.model import Seq2SeqModel
from img2latex_amd.training.predictor import Predictor, TokenTable
dev = torch.device("cuda:0")
cfg = synth.model_config()
model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0, enc_scale=16.0).items()})
model = model.to(dev).eval()
B, T = 256, 150
vocab = {"<PAD>": 0, "<START>": 1, "<END>": 2, "<UNK>": 3}
vocab.update({f"t{i}": i for i in range(4, cfg["vocab_size"])})
pred = Predictor(model, TokenTable(vocab, max_sequence_length=T), device=dev)
sizes = [(30 + (7 * k) % 90, 80 + (53 * k) % 700, 1 + 2 * (k % 2)) for k in range(B)]
pages = []
for k, (h, w, c) in enumerate(sizes):
    base = synth.uniform(5000 + k, "img", (h, w, c), 0.0, 255.0)
    pages.append(np.round(base).astype(np.uint8).reshape((h, w) if c == 1 else (h, w, 3)))
tg = torch.from_numpy(synth.make_formulas(B, T, cfg["vocab_size"], seed=777)).to(dev)
list(pred.evaluate_stream([(pages, tg)] * 5, max_length=T))
torch.cuda.synchronize()
for la in (1, 2, 1, 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    list(pred.evaluate_stream([(pages, tg)] * 40, max_length=T, lookahead=la))
    print(f"lookahead {la}: {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms per batch")
