"""VERDICT r03 item 4, the micro-experiment: decode-shaped dummy kernels whose per-step products run as a burst of
v_mfma_f32_16x16x32_bf16 (split-bf16, 6 partial products) against the r03 dummy's v_pk_fma_f32 burst -- alone, and beside
the real conv encoder on a second stream.  Per variant: step time alone, the burst's own duration (in-kernel clock, alone and
beside the encoder), the encoder's time beside it and the pair's time.  Burst sizes per wave and step:
   FMA  8 members x 8 rows : 768 packed FMAs per lane (the r03 figure: 8 x 256 x (128 + 64) MACs per member)
   MFMA 8 members x 8 rows : 16 x (128 + 64) x 256 x 6 products / (16 x 16 x 32) / 4 waves = 144   (half of M is padding)
   MFMA 16 members x 16 rows: 16 x (64 + 32) x 256 x 6 / 8192 / 4 = 72
   MFMA 8 members x 16 rows (128 CUs): 16 x 192 x 256 x 6 / 8192 / 4 = 144 on HALF the workgroups"""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "hmer-img2latex_amd"))
from img2latex_amd import synth
from img2latex_amd.model import Seq2SeqModel
so = os.path.join(HERE, "libcoresident_mfma.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "coresident_mfma.hip")], check=True)
lib = ctypes.CDLL(so)
lib.launch_fake.argtypes = [ctypes.c_int] * 6 + [ctypes.c_void_p] * 3
dev = torch.device("cuda:0")
cfg = synth.model_config()
m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg)).to(dev).eval()
x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
sink = torch.zeros(256, device=dev)
ticks = torch.zeros(256, dtype=torch.int64, device=dev)
s_dec, s_enc = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev)
STEPS = 150

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
variants = (("FMA burst 768/lane, 256 workgroups (r03 dummy)", 0, 256, 768, 2600),
            ("MFMA burst 144/wave, 256 workgroups (8 members x 8 rows)", 1, 256, 144, 2600),
            ("MFMA burst 72/wave, 256 workgroups (16 members x 16 rows)", 1, 256, 72, 2600),
            ("MFMA burst 144/wave, 128 workgroups (8 members x 16 rows)", 1, 128, 144, 2600),
            ("MFMA burst 288/wave, 128 workgroups (4 members x 8 rows... x2 sets)", 1, 128, 288, 2600),
            ("no burst, idle only, 256 workgroups", 1, 256, 0, 4100))
for name, kind, wgs, ops, idle in variants:
    def fake():
        assert lib.launch_fake(kind, wgs, STEPS, ops, idle, 85 * 1024, sink.data_ptr(), ticks.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    enc_t = [0.0]
    def both():
        main = torch.cuda.current_stream()
        s_dec.wait_stream(main); s_enc.wait_stream(main)
        with torch.cuda.stream(s_dec): fake()
        with torch.cuda.stream(s_enc), torch.no_grad():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); m.encoder(x); b.record()
        main.wait_stream(s_dec); main.wait_stream(s_enc)
        enc_t[0] = (a, b)
    alone = timed(fake)
    burst_alone = float(ticks[:wgs].double().mean()) / STEPS / 100.0           # us per step
    tg = timed(both)
    burst_beside = float(ticks[:wgs].double().mean()) / STEPS / 100.0
    torch.cuda.synchronize()
    e_beside = enc_t[0][0].elapsed_time(enc_t[0][1])
    print(f"{name:68s}: alone {alone:.3f} ms (burst {burst_alone:.2f} us/step), pair {tg:.3f} ms (burst {burst_beside:.2f} us/step, "
          f"encoder {e_beside:.3f} ms = x{e_beside / enc_alone:.2f}); sum {alone + enc_alone:.3f}, max {max(alone, enc_alone):.3f}")
