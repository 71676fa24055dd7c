"""ResNet-50 trunk at B=256: one launch chain over the whole batch against the batch cut into 2 / 4 slices whose chains run
side by side on separate HIP streams (each slice's launch gaps, tile tails and under-filled layers are filled by the others)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "hmer-img2latex_amd"))
from img2latex_amd import _lib, synth
from img2latex_amd.model import Seq2SeqModel
dev = torch.device("cuda:0")
cfg = synth.model_config()
enc_p = dict(img_height=64, img_width=320, channels=3, model_name="resnet50", embedding_dim=256, freeze_backbone=True)
model = Seq2SeqModel("resnet_lstm", cfg["vocab_size"], enc_p, synth.decoder_params(cfg))
shapes = [(k, tuple(v.shape)) for k, v in model.encoder.state_dict().items()]
full = {"encoder." + k: torch.from_numpy(v) for k, v in synth.make_resnet_state_dict(shapes, seed=5).items()}
full.update({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0).items() if k.startswith("decoder.")})
model.load_state_dict(full)
model = model.to(dev).eval()
enc = model.encoder
x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
N = 100
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]

def whole():
    return enc.trunk(x)

def split(n):
    cur = torch.cuda.current_stream()
    parts = []
    step = 256 // n
    for i in range(n):
        s = streams[i]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            parts.append(enc.trunk(x[i * step:(i + 1) * step]))
    for i in range(n):
        cur.wait_stream(streams[i])
    return torch.cat(parts)

def batches_side_by_side(n):          # n whole batches in flight (what a pipeline with n encoder streams does)
    cur = torch.cuda.current_stream()
    outs = []
    for i in range(n):
        s = streams[i]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(enc.trunk(x))
    for i in range(n):
        cur.wait_stream(streams[i])
    return outs[0]

def timeit(fn, per=1):
    with torch.no_grad():
        for _ in range(10):
            out = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N):
            out = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N / per * 1e3, out

t_whole, ref = timeit(whole)
print(f"whole batch, one stream: {t_whole:.3f} ms per 256 images")
for n in (2, 4):
    t, out = timeit(lambda: split(n))
    print(f"{n} slices on {n} streams: {t:.3f} ms per 256 images, max |diff| vs whole {float((out - ref).abs().max()):.3g}")
for n in (2, 3):
    t, out = timeit(lambda: batches_side_by_side(n), per=n)
    print(f"{n} whole batches side by side: {t:.3f} ms per 256 images")
