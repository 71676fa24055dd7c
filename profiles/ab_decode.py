"""A/B timing of the greedy decode loop with a given build of the library (arg 1 = path of the .so, default = product):
same box, same process layout.  usage: python profiles/ab_decode.py [lib.so] ; prints ms per decode launch (median of 40)."""
import os, sys, statistics, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "hmer-img2latex_amd"))
from img2latex_amd import synth, _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = sys.argv[1]
from img2latex_amd.model import Seq2SeqModel
cfg = synth.model_config()
dev = torch.device("cuda:0")
model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0, enc_scale=16.0).items()})
model = model.to(dev).eval()
images = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
with torch.no_grad():
    enc = model.encoder(images)
    for _ in range(5):
        ids, _ = model.greedy_ids(enc, synth.START, synth.END, 150)
    ts = []
    for _ in range(40):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ids, _ = model.greedy_ids(enc, synth.START, synth.END, 150); b.record()
        torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "product", "prepare+decode ms: median %.4f min %.4f" % (statistics.median(ts), min(ts)),
      "ids checksum", int(ids.sum()))
