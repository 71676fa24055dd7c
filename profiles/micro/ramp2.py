"""After 150 continuous pipelined batches: drain + synchronize (+ an idle gap), then per-batch times of the next 30 batches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "hmer-img2latex_amd"))
from img2latex_amd import _lib, synth
from img2latex_amd.model import Seq2SeqModel
from img2latex_amd.pipeline import GreedyPipeline
dev = torch.device("cuda:0")
cfg = synth.model_config()
model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0, enc_scale=16.0).items()})
model = model.to(dev).eval()
x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
pipe = GreedyPipeline(model, synth.START, synth.END, 150, rows_per_workgroup=0, decode_flags=(_lib.FLAG_DECODE_GROUP8 if os.environ.get("MEMBERS") == "8" else _lib.FLAG_DECODE_GROUP16), decode_priority=-1, hold_encoder=(os.environ['HOLD'] != '0' if 'HOLD' in os.environ else None))   # HOLD=0: no residency dependency (A/B)
def run(n, stamps=None):
    t0 = time.perf_counter()
    for i in range(n):
        if pipe.pending() >= pipe.depth:
            pipe.collect()
            if stamps is not None: stamps.append(time.perf_counter() - t0)
        pipe.submit(x)
    while pipe.pending():
        pipe.collect()
        if stamps is not None: stamps.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run(150)
for gap_ms in (0.0, 1.0, 5.0):
    run(60)
    time.sleep(gap_ms / 1e3)
    st = []
    tot = run(30, st)
    d = [st[0]] + [b - a for a, b in zip(st, st[1:])]
    print(f"gap {gap_ms:5.1f} ms: 30 batches in {tot * 1e3:.2f} ms ({tot / 30 * 1e3:.3f} per batch):", " ".join(f"{v * 1e3:.2f}" for v in d))
