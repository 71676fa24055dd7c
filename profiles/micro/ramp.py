"""Per-batch completion times of the first batches after start-up (headline workload): where is the start-up ramp?"""
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
mode = sys.argv[1] if len(sys.argv) > 1 else "pipe"
N = 80
stamps = []
res = []
torch.cuda.synchronize()
t0 = time.perf_counter()
if mode == "pipe":
    pipe = GreedyPipeline(model, synth.START, synth.END, 150, rows_per_workgroup=0, decode_flags=_lib.FLAG_DECODE_GROUP8, decode_priority=-1)
    for i in range(N):
        if pipe.pending() >= pipe.depth:
            pipe.collect(); stamps.append(time.perf_counter() - t0); res.append(torch.cuda.memory_reserved() >> 20)
        pipe.submit(x)
    while pipe.pending():
        pipe.collect(); stamps.append(time.perf_counter() - t0)
else:
    host = torch.empty((256, 150), dtype=torch.int32).pin_memory()
    for i in range(N):
        with torch.no_grad():
            ids, _ = model.greedy_ids(model.encoder(x), synth.START, synth.END, 150)
            host.copy_(ids, non_blocking=True)
        torch.cuda.current_stream().synchronize(); stamps.append(time.perf_counter() - t0)
d = [stamps[0]] + [b - a for a, b in zip(stamps, stamps[1:])]
print("reserved MiB after each collect:", " ".join(str(v) for v in res))
print(mode, "ms per batch:", " ".join(f"{v * 1e3:.2f}" for v in d))
