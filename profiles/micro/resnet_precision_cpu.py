"""CPU experiment (no GPU): which storage roundings of the ResNet TRAINING data path cost how much end to end?
The fp32 restatement (oracle/resnet_oracle.py) is re-run with mantissa rounding injected at chosen points
(conv operands x / w, raw conv output z, normalised value, block activation y) and compared with plain fp32 (and
fp64 to show the net's own noise amplification).  Output: output error and gradient cosines per variant.
usage: python profiles/micro/resnet_precision_cpu.py [resnet50] [B] [H] [W]"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "hmer-img2latex_amd")); sys.path.insert(0, os.path.join(R, "oracle"))
import resnet_oracle as RO
from img2latex_amd import synth
from img2latex_amd.model import ResNetEncoder


def rnd(t, bits):
    """round to `bits` mantissa bits (8 = bf16, 11 = fp16-like, 16 = two bf16 pieces), straight-through gradient"""
    if bits is None or bits >= 24:
        return t
    d = t.detach()
    if bits == 8:
        q = d.float().to(torch.bfloat16).to(d.dtype)
    else:
        m, e = torch.frexp(d.double())
        q = torch.ldexp(torch.round(m * 2.0 ** bits) / 2.0 ** bits, e).to(d.dtype)
    return t + (q - d)


def trunk(sd, name, x, P, prefix="encoder.resnet."):
    kind, counts = RO.BLOCKS[name]
    def bn(key, z):
        return rnd(F.batch_norm(z, None, None, sd[key + ".weight"], sd[key + ".bias"], training=True, eps=1e-5), P.get("yn"))
    conv = lambda t, w, **kw: rnd(F.conv2d(rnd(t, P.get("x")), rnd(w, P.get("w")), **kw), P.get("z"))
    x = F.relu(bn(prefix + "1", conv(x, sd[prefix + "0.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    for li, n in enumerate(counts):
        for bi in range(n):
            p = f"{prefix}{4 + li}.{bi}."
            stride = 2 if (li > 0 and bi == 0) else 1
            identity = x
            if kind == "bottleneck":
                o = F.relu(bn(p + "bn1", conv(x, sd[p + "conv1.weight"])))
                o = F.relu(bn(p + "bn2", conv(o, sd[p + "conv2.weight"], stride=stride, padding=1)))
                o = bn(p + "bn3", conv(o, sd[p + "conv3.weight"]))
            else:
                o = F.relu(bn(p + "bn1", conv(x, sd[p + "conv1.weight"], stride=stride, padding=1)))
                o = bn(p + "bn2", conv(o, sd[p + "conv2.weight"], padding=1))
            if p + "downsample.0.weight" in sd:
                identity = bn(p + "downsample.1", conv(x, sd[p + "downsample.0.weight"], stride=stride))
            x = rnd(F.relu(o + identity), P.get("y"))
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


def step(sd0, name, x, dout, trainable, P, dtype=torch.float32):
    sd = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    for n in trainable:
        sd[n].requires_grad_(True)
    feat = trunk(sd, name, x.to(dtype), P)
    out = F.relu(F.linear(feat, sd["encoder.embedding_layer.weight"], sd["encoder.embedding_layer.bias"]))
    (out * dout.to(dtype)).sum().backward()
    return out.detach().double(), {n: sd[n].grad.detach().double() for n in trainable}


name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
B, H, W = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 4), (3, 64), (4, 320)))
enc = ResNetEncoder(H, W, 3, model_name=name, embedding_dim=64, freeze_backbone=False)
shapes = [(k, tuple(v.shape)) for k, v in enc.state_dict().items()]
np_sd = synth.make_resnet_state_dict(shapes, seed=11)
sd = {"encoder." + k: torch.from_numpy(v.copy()) for k, v in np_sd.items()}
trainable = ["encoder." + n for n, p in enc.named_parameters()]
x = torch.from_numpy(synth.uniform(5, "images", (B, 3, H, W), -1.0, 1.0))
dout = torch.from_numpy(synth.uniform(6, "dout", (B, 64), -1.0, 1.0))
torch.set_num_threads(8)
ref_out, ref_g = step(sd, name, x, dout, trainable, {})
variants = {
    "fp64 (noise floor of fp32 itself)": (dict(), torch.float64),
    "r03: x w z yn y all bf16": (dict(x=8, w=8, z=8, yn=8, y=8), torch.float32),
    "z fp32; x w yn y bf16": (dict(x=8, w=8, yn=8, y=8), torch.float32),
    "z fp32, yn fp32; x w y bf16": (dict(x=8, w=8, y=8), torch.float32),
    "only conv operands bf16 (x, w)": (dict(x=8, w=8), torch.float32),
    "only w bf16": (dict(w=8), torch.float32),
    "only x bf16": (dict(x=8), torch.float32),
    "operands 11 bits (fp16-like)": (dict(x=11, w=11), torch.float32),
    "operands 16 bits (2 x bf16)": (dict(x=16, w=16), torch.float32),
    "x 16 bits, w 8 bits": (dict(x=16, w=8), torch.float32),
}
for label, (P, dt) in variants.items():
    out, g = step(sd, name, x, dout, trainable, P, dt)
    e = float((out - ref_out).abs().max()) / max(1.0, float(ref_out.abs().max()))
    cs = np.sort([float((g[n].flatten() @ ref_g[n].flatten()) / (g[n].norm() * ref_g[n].norm() + 1e-300)) for n in trainable])
    print(f"{label:40s} out err {e:.3e}   cos min {cs[0]:.4f}  p5 {cs[len(cs) // 20]:.4f}  median {np.median(cs):.4f}", flush=True)
