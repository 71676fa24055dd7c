"""Host time to enqueue one ResNet-50 trunk + FC (53 conv launches from Python) against its GPU time."""
import os, sys, time
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(R, "hmer-img2latex_amd"))
from img2latex_amd import synth
from img2latex_amd.model import ResNetEncoder

dev = torch.device("cuda:0")
enc = ResNetEncoder(64, 320, 3, model_name="resnet50", embedding_dim=256)
shapes = [(k, tuple(v.shape)) for k, v in enc.state_dict().items()]
enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_resnet_state_dict(shapes, seed=5).items()})
enc = enc.to(dev).eval()
x = torch.from_numpy(synth.make_images(256, synth.model_config(), seed=3)).to(dev)
with torch.no_grad():
    for _ in range(5):
        enc(x)
    torch.cuda.synchronize()
    for n in (1, 10, 30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            enc(x)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        ta = time.perf_counter() - t0
        print(f"{n:3d} trunks: host enqueue {th / n * 1e3:.3f} ms each, with the final synchronise {ta / n * 1e3:.3f} ms each", flush=True)
