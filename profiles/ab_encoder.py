"""A/B timing of the CNN encoder's launches with a given build of the library (arg 1 = path of the .so, default = product;
arg 2 = batch, default 256; arg 3 = 'train' to time the training forward with arg max).  Same box, same process layout."""
import os, sys, statistics, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "hmer-img2latex_amd"))
from img2latex_amd import synth, _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
train = len(sys.argv) > 3 and sys.argv[3] == "train"
from img2latex_amd.model import Seq2SeqModel
cfg = synth.model_config()
dev = torch.device("cuda:0")
model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg)).to(dev).eval()
images = torch.from_numpy(synth.make_images(B, cfg, seed=1234)).to(dev)
marks = []
_lib.set_stage_hook(lambda name: marks.append((name, torch.cuda.Event(enable_timing=True))) or marks[-1][1].record())
acc = {}
with torch.no_grad():
    for it in range(35):
        marks.clear()
        am = [] if train else None
        model.encoder.conv_blocks(images, am)
        torch.cuda.synchronize()
        if it >= 5:
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                acc.setdefault(n1, []).append(e0.elapsed_time(e1))
print(os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "product", f"B={B}", "train" if train else "inference",
      " ".join(f"{k}={statistics.median(v) * 1e3:.1f}us" for k, v in acc.items()))
