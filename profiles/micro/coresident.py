"""Driver of coresident.hip: the CNN encoder of one batch (the product's kernels) beside a decode-shaped kernel of the
previous batch on two streams, against each of them alone.  Prints milliseconds."""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "hmer-img2latex_amd"))
from img2latex_amd import synth
from img2latex_amd.model import Seq2SeqModel

so = os.path.join(HERE, "libcoresident.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so,
                    os.path.join(HERE, "coresident.hip")], check=True)
lib = ctypes.CDLL(so)
lib.launch_fake_decode.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
cfg = synth.model_config()
m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg)).to(dev).eval()
x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
sink = torch.zeros(256, device=dev)
s_dec, s_enc = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def fake(steps=150, fmas=768, idle_ns=2600, lds=85 * 1024, wgs=256):
    rc = lib.launch_fake_decode(wgs, steps, fmas, idle_ns, lds, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def timed(fn, n=20):
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


def both():
    main = torch.cuda.current_stream()
    s_dec.wait_stream(main); s_enc.wait_stream(main)
    with torch.cuda.stream(s_dec):
        fake()
    with torch.cuda.stream(s_enc), torch.no_grad():
        m.encoder(x)
    main.wait_stream(s_dec); main.wait_stream(s_enc)


with torch.no_grad():
    enc_alone = timed(lambda: m.encoder(x))
dec_alone = timed(fake)
together = timed(both)
print(f"encoder alone {enc_alone:.3f} ms, decode-shaped kernel alone {dec_alone:.3f} ms (150 steps: {dec_alone / 150 * 1e3:.2f} us per step), "
      f"both on two streams {together:.3f} ms  (sum {enc_alone + dec_alone:.3f} ms, max {max(enc_alone, dec_alone):.3f} ms)")
for lds in (85, 140):
    dec = timed(lambda: fake(lds=lds * 1024))
    with torch.no_grad():
        tg = timed(lambda: (fake(lds=lds * 1024), None) and None)
    print(f"  [decode-shaped kernel with {lds} KB of LDS alone: {dec:.3f} ms]")
