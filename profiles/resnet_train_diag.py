"""Where do the HIP training forward and the fp32 oracle part ways (r03: the bf16-emulating oracle)?  Per unit: max |diff| / max |ref| of the
activation, the smallest per-channel std / |mean| of the raw conv output (a channel whose spread is below bf16's
resolution of its mean is normalised to rounding noise)."""
import os, sys
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "hmer-img2latex_amd")); sys.path.insert(0, os.path.join(R, "oracle"))
import resnet_oracle as RO
from img2latex_amd import synth
from img2latex_amd.model import ResNetEncoder
from img2latex_amd.model._train_fn import encoder_train_forward

name, B, H, W = sys.argv[1] if len(sys.argv) > 1 else "resnet50", int(sys.argv[2]) if len(sys.argv) > 2 else 4, 64, 320
enc = ResNetEncoder(H, W, 3, model_name=name, embedding_dim=64)
shapes = [(k, tuple(v.shape)) for k, v in enc.state_dict().items()]
np_sd = synth.make_resnet_state_dict(shapes, seed=11)
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in np_sd.items()})
enc = enc.cuda().train()
sd = {"encoder." + k: torch.from_numpy(v.copy()) for k, v in np_sd.items()}
x = torch.from_numpy(synth.uniform(5, "images", (B, 3, H, W), -1.0, 1.0))
out, tape = encoder_train_forward(enc, x.cuda())
taps = {}
RO.resnet_trunk_train(sd, name, x, {}, emulate_bf16=False, taps=taps)
names = {id(m): n for n, m in enc.named_modules()}
for u in tape["units"]:
    key = "encoder." + names[id(u["conv"])]
    ref = taps[key].permute(0, 2, 3, 1)
    got = u["y"].float().cpu()
    z = u["z"].float()
    ratio = (z.std(dim=(0, 1, 2)) / (z.mean(dim=(0, 1, 2)).abs() + 1e-12)).min()
    print(f"{key:38s} rel diff {float((got - ref).abs().max()) / float(ref.abs().max()):.3e}   min std/|mean| of z {float(ratio):.3e}"
          f"   M {z.numel() // z.shape[-1]}")
