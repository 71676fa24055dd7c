"""ResNet-50 trunk (B=256, 64x320) with and without the bottleneck joins: features bit-identical, time per trunk."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "hmer-img2latex_amd"))
from img2latex_amd.model import ResNetEncoder
from img2latex_amd import synth

dev = torch.device("cuda:0")
enc = ResNetEncoder(64, 320, 3, model_name="resnet50", embedding_dim=256)
shapes = [(k, tuple(v.shape)) for k, v in enc.state_dict().items()]
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_resnet_state_dict(shapes, seed=5).items()})
enc = enc.to(dev).eval()
x = torch.from_numpy(synth.make_images(256, synth.model_config(), seed=3)).to(dev)
feats = {}
with torch.no_grad():
    for fuse in (True, False, True, False):
        enc.fuse_joins = fuse
        for _ in range(5):
            f = enc.trunk(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            f = enc.trunk(x)
        e1.record()
        torch.cuda.synchronize()
        feats[fuse] = f.clone()
        print(f"fuse_joins={fuse}: {e0.elapsed_time(e1) / 30:.4f} ms per trunk", flush=True)
print("features bit-identical:", torch.equal(feats[True], feats[False]))
