"""Which resource costs the conv encoder its speed beside a decode-shaped kernel?  Variants of fake_decode."""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "hmer-img2latex_amd"))
from img2latex_amd import synth
from img2latex_amd.model import Seq2SeqModel
so = os.path.join(HERE, "libcoresident.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "coresident.hip")], check=True)
lib = ctypes.CDLL(so)
lib.launch_fake_decode.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = synth.model_config()
m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg)).to(dev).eval()
x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
sink = torch.zeros(256, device=dev)
s_dec, s_enc = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev)

def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

with torch.no_grad():
    enc_alone = timed(lambda: m.encoder(x))
print(f"encoder alone {enc_alone:.3f} ms")
for name, kw in (("FMA bursts + idle, 85 KB LDS", dict(fmas=768, idle_ns=2600, lds=85 * 1024)),
                 ("idle only, 85 KB LDS (occupancy alone)", dict(fmas=0, idle_ns=4100, lds=85 * 1024)),
                 ("FMA bursts + idle, 16 KB LDS (two conv workgroups still fit)", dict(fmas=768, idle_ns=2600, lds=16 * 1024)),
                 ("idle only, 16 KB LDS", dict(fmas=0, idle_ns=4100, lds=16 * 1024))):
    def fake():
        assert lib.launch_fake_decode(256, 150, kw["fmas"], kw["idle_ns"], kw["lds"], sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    def both():
        main = torch.cuda.current_stream()
        s_dec.wait_stream(main); s_enc.wait_stream(main)
        with torch.cuda.stream(s_dec): fake()
        with torch.cuda.stream(s_enc), torch.no_grad(): m.encoder(x)
        main.wait_stream(s_dec); main.wait_stream(s_enc)
    alone = timed(fake)
    tg = timed(both)
    print(f"{name:62s}: alone {alone:.3f} ms, beside the encoder {tg:.3f} ms (sum {alone + enc_alone:.3f}, max {max(alone, enc_alone):.3f})")
