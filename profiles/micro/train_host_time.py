"""Is the training step bound by the host?  Host time to ENQUEUE one step (no synchronisation) against the step's wall time."""
import os, sys, time
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "hmer-img2latex_amd"))
from img2latex_amd import synth
from img2latex_amd.model import Seq2SeqModel
from img2latex_amd.training import TrainStep

dev = torch.device("cuda:0")
cfg = synth.model_config(dropout=0.1)
model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42).items()})
model = model.to(dev).train()
ts = TrainStep(model, seed=1)
images = torch.from_numpy(synth.make_images(64, cfg, seed=1234)).to(dev)
forms = torch.from_numpy(synth.make_formulas(64, 150, cfg["vocab_size"], seed=99)).to(dev)


def step():
    ts.forward_backward(images, forms)
    ts.apply()


for _ in range(30):
    step()
torch.cuda.synchronize()
for n in (1, 5, 40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{n:3d} steps: host enqueue {t_host / n * 1e3:.3f} ms per step, with the final synchronise {t_all / n * 1e3:.3f} ms per step", flush=True)
