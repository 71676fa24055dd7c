"""Diagnostic: how long the pixel upload and the encoder/decode take when they overlap (events on both streams)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hmer-img2latex_amd"))
from img2latex_amd import synth
from img2latex_amd.model import Seq2SeqModel
cfg = synth.model_config()
dev = torch.device("cuda:0")
model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg)).to(dev).eval()
B, T = 256, 150
images = torch.rand(B, 3, 64, 320, device=dev) * 2 - 1
host = images.cpu().pin_memory()
bufs = [torch.empty_like(images) for _ in range(3)]
copy_s, main_s = torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
E = lambda: torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    for mode in ("alone", "overlap"):
        for _ in range(3):
            enc = model.encoder(bufs[0]); model.greedy_ids(enc, 1, 2, T)
        torch.cuda.synchronize()
        n = 10
        cu, cc = [(E(), E()) for _ in range(n)], [(E(), E()) for _ in range(n)]
        t0 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(copy_s):
                cu[i][0].record(copy_s)
                bufs[1 + (i & 1)].copy_(host, non_blocking=True)
                cu[i][1].record(copy_s)
            if mode == "alone":
                torch.cuda.synchronize()
            cc[i][0].record(main_s)
            enc = model.encoder(bufs[0]); model.greedy_ids(enc, 1, 2, T)
            cc[i][1].record(main_s)
            if mode == "alone":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(mode, "wall/iter %.3f ms" % (dt * 1e3), "upload %.3f ms" % (sum(a.elapsed_time(b) for a, b in cu) / n),
              "compute %.3f ms" % (sum(a.elapsed_time(b) for a, b in cc) / n))

def chained(nbuf, n=12, d2h=None):
    uploaded = [torch.cuda.Event() for _ in range(nbuf)]
    consumed = [torch.cuda.Event() for _ in range(nbuf)]
    cu, cc = [(E(), E()) for _ in range(n)], [(E(), E()) for _ in range(n)]
    def upload(i):
        with torch.cuda.stream(copy_s):
            copy_s.wait_event(consumed[i % nbuf])
            cu[i][0].record(copy_s)
            bufs[i % nbuf].copy_(host, non_blocking=True)
            cu[i][1].record(copy_s)
            uploaded[i % nbuf].record(copy_s)
    for e in consumed:
        e.record(main_s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(min(nbuf - 1, n)):
        upload(j)
    for i in range(n):
        if i + nbuf - 1 < n:
            upload(i + nbuf - 1)
        main_s.wait_event(uploaded[i % nbuf])
        cc[i][0].record(main_s)
        enc = model.encoder(bufs[i % nbuf]); ids, _ = model.greedy_ids(enc, 1, 2, T)
        cc[i][1].record(main_s)
        consumed[i % nbuf].record(main_s)
        if d2h == "main":
            ids_h[i & 1].copy_(ids, non_blocking=True)
        elif d2h == "late":
            if i > 0:
                ids_h[(i - 1) & 1].copy_(prev_ids, non_blocking=True)
            prev_ids = ids
        elif d2h == "third":
            with torch.cuda.stream(down_s):
                down_s.wait_event(consumed[i % nbuf])
                ids_h[i & 1].copy_(ids, non_blocking=True)
                ids.record_stream(down_s)
    host_done = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("chained nbuf", nbuf, "d2h", d2h, "wall/iter %.3f ms" % (dt * 1e3), "host issue %.3f ms/iter" % (host_done / n * 1e3),
          "upload %.3f" % (sum(a.elapsed_time(b) for a, b in cu) / n), "compute %.3f" % (sum(a.elapsed_time(b) for a, b in cc) / n))
    base = cc[0][0]
    print("  compute starts:", " ".join("%.2f" % base.elapsed_time(a) for a, _ in cc))
    print("  upload  starts:", " ".join("%.2f" % base.elapsed_time(a) for a, _ in cu))
with torch.no_grad():
    ids_h = [torch.empty(B, T, dtype=torch.int32).pin_memory() for _ in range(2)]
    down_s = torch.cuda.Stream(dev)
    chained(2, d2h="late"); chained(2, d2h="late"); chained(2, n=40, d2h="late")
