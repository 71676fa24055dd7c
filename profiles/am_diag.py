"""Diagnostic: pooling arg max / ReLU gate of the default training forward vs the exact-fp32 kernels, primary fixture."""
import os, sys, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "hmer-img2latex_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from helpers import images, load, torch_state_dict
from img2latex_amd import _lib
from img2latex_amd.model import Seq2SeqModel
import torch.nn.functional as F
from test_hip_training import build
d, cfg, m = build("primary")
from img2latex_amd import synth
import numpy as np
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
imgs = images(cfg) if NB == 4 else torch.from_numpy(synth.make_images(NB, cfg, seed=1234))
x = imgs.cuda()
res = {}
for name, fl in (("default", 0), ("exact", _lib.FLAG_EXACT_FP32)):
    m.encoder.kernel_flags = fl
    am = []
    ys = m.encoder.conv_blocks(x, am)
    res[name] = ([y.cpu() for y in ys], [a.cpu() for a in am])
sd = torch_state_dict("primary")
inp = imgs.double()
for i in range(3):
    yd, ad = res["default"][0][i], res["default"][1][i]
    ye, ae = res["exact"][0][i], res["exact"][1][i]
    w, b = sd[f"encoder.cnn_layers.{3*i}.weight"].double(), sd[f"encoder.cnn_layers.{3*i}.bias"].double()
    conv = F.conv2d(inp, w, None, padding=1)
    B_, C_, Hc, Wc = conv.shape
    quads = conv.reshape(B_, C_, Hc // 2, 2, Wc // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(B_, C_, Hc // 2, Wc // 2, 4)
    top2 = quads.topk(2, dim=-1).values
    gap = top2[..., 0] - top2[..., 1]
    pre = top2[..., 0] + b[None, :, None, None]
    mag = quads.abs().amax(-1).clamp_min(1e-30)
    live = pre > 0
    dif_am = (ad != ae) & live
    dif_gate = (yd > 0) != (ye > 0)
    print(f"block {i}: windows {ad.numel()}  argmax differs (live) {int(dif_am.sum())}  gate differs {int(dif_gate.sum())}  max|y diff| {float((yd-ye).abs().max()):.3e}")
    for nm, a_, y_ in (("default", ad, yd), ("exact", ae, ye)):
        wrong = (a_.long() != quads.argmax(-1)) & live & (gap > 0)
        wg = (y_ > 0) != (pre > 0)
        print(f"   {nm}: argmax off fp64 {int(wrong.sum())} (max gap/mag {float((gap[wrong]/mag[wrong]).max()) if wrong.any() else 0:.2e})  gate off {int(wg.sum())}")
        for idx in wrong.nonzero()[:4].tolist() + wg.nonzero()[:4].tolist():
            q = quads[tuple(idx)]
            print("      window", idx, "quad", [f"{v:.9e}" for v in q.tolist()], "bias", float(b[idx[1]]), "pre", float(pre[tuple(idx)]), "am", int(a_[tuple(idx)]), "y", float(y_[tuple(idx)]))
    inp = torch.relu(F.max_pool2d(conv + b[None, :, None, None], 2))
