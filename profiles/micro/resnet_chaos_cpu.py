"""CPU control for tests/test_resnet_training.py::test_resnet50_lstm_30_step_loss_curve_vs_fp32_oracle: how far do two
CORRECT evaluations of the same 30 optimisation steps drift apart?  The oracle's steps in float32 against the same steps
in float64 (a perturbation of ~6e-8 per operation): worst relative loss difference and running-statistic distance."""
import os, sys
import numpy as np
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "hmer-img2latex_amd")); sys.path.insert(0, os.path.join(R, "oracle"))
import resnet_oracle as RO
from img2latex_amd import synth
from img2latex_amd.model import Seq2SeqModel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cfg = synth.model_config(vocab_size=60, embedding_dim=64, hidden_dim=64, dropout=0.0)
enc_p = dict(img_height=64, img_width=320, channels=3, model_name="resnet50", embedding_dim=64, freeze_backbone=False)
m = Seq2SeqModel("resnet_lstm", 60, enc_p, synth.decoder_params(cfg))
shapes = [(k, tuple(v.shape)) for k, v in m.encoder.state_dict().items()]
full = {"encoder." + k: torch.from_numpy(v) for k, v in synth.make_resnet_state_dict(shapes, seed=3).items()}
full.update({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=4).items() if k.startswith("decoder.")})
trainable = {n for n, p in m.named_parameters()}
B = 8
batches = [(torch.from_numpy(synth.uniform(20 + i, "images", (B, 3, 64, 320), -1.0, 1.0)),
            torch.from_numpy(synth.make_formulas(B, 14, 60, seed=30 + i, min_len=5))) for i in range(4)]
runs = {}
for dt in (torch.float32, torch.float64):
    sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in full.items()}
    state, losses = {}, []
    torch.set_default_dtype(dt)
    for i in range(steps):
        x, f = batches[i % 4]
        losses.append(RO.resnet_lstm_train_step(sd, "resnet50", cfg, x.to(dt), f, state, trainable, lr=1e-3)["loss"])
    runs[dt] = (np.array(losses), sd)
a, b = runs[torch.float32], runs[torch.float64]
rel = np.abs(a[0] - b[0]) / b[0]
print("losses fp32", np.round(a[0], 4))
print("rel diff per step", np.array2string(rel, precision=2))
rs = max(float((a[1][k].double() - b[1][k]).abs().max() / b[1][k].abs().max()) for k in full if "running" in k)
print(f"worst loss rel diff {rel.max():.3e}; running statistics rel to max {rs:.3e}")
