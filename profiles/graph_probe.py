"""Does capturing the launch chains in a HIP graph pay?  ResNet-50 trunk (55 launches, 20-50 us each) and one training
step (65 launches): eager back-to-back launches vs replay of a captured graph, same kernels, same buffers."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hmer-img2latex_amd"))
from img2latex_amd import synth
from img2latex_amd.model import ResNetEncoder, Seq2SeqModel

dev = torch.device("cuda:0")
cfg = synth.model_config()
enc = ResNetEncoder(64, 320, 3, model_name="resnet50", embedding_dim=256)
shapes = [(k, tuple(v.shape)) for k, v in enc.state_dict().items()]
enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_resnet_state_dict(shapes, seed=5).items()})
enc = enc.to(dev).eval()
x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    eager = timeit(lambda: enc(x))
    want = enc(x).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            enc(x)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = enc(x)
    graph = timeit(g.replay)
    g.replay()
    torch.cuda.synchronize()
    print(f"resnet50 encoder B=256: eager {eager:.3f} ms, graph replay {graph:.3f} ms, identical {torch.equal(out, want)}")
