"""Soak of GreedyPipeline as bench.py uses it (r04: encoder held behind the decode's residency signal, decoder prepare on the encoder stream
into rotating workspaces): thousands of batches from three alternating image sets, EVERY result compared with the
one-batch-at-a-time search of the same kernel; throughput of the run."""
import os, sys, time
import numpy as np
import torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "hmer-img2latex_amd"))
from img2latex_amd import _lib, synth
from img2latex_amd.model import Seq2SeqModel
from img2latex_amd.pipeline import GreedyPipeline
dev = torch.device("cuda:0")
cfg = synth.model_config()
FLAG = _lib.FLAG_DECODE_GROUP8 if os.environ.get("MEMBERS") == "8" else _lib.FLAG_DECODE_GROUP16    # r04 default: the matrix-core decode

def soak(model, name, n, enc_streams):
    sets = [torch.from_numpy(synth.make_images(256, cfg, seed=s)).to(dev) for s in (1234, 77, 78)]
    with torch.no_grad():
        want = [_lib.check_ids(model.greedy_ids(model.encoder(x), synth.START, synth.END, 150, flags=FLAG)[0].cpu()).numpy().copy() for x in sets]
    pipe = GreedyPipeline(model, synth.START, synth.END, 150, rows_per_workgroup=0, decode_flags=FLAG,
                          decode_priority=-1, encoder_streams=enc_streams)
    order, got_bad, k = [], 0, 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if pipe.pending() >= pipe.depth:
            g = pipe.collect().numpy()
            got_bad += int(not np.array_equal(g, want[order[k]])); k += 1
        j = (i * 7 + i // 5) % 3
        order.append(j)
        pipe.submit(sets[j])
        if i % 997 == 996:                      # a fence now and then: the pipeline restarts from empty
            while pipe.pending():
                g = pipe.collect().numpy()
                got_bad += int(not np.array_equal(g, want[order[k]])); k += 1
            torch.cuda.synchronize()
    while pipe.pending():
        g = pipe.collect().numpy()
        got_bad += int(not np.array_equal(g, want[order[k]])); k += 1
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {n} batches ({k} compared), {got_bad} differ from the one-at-a-time search, {n * 256 * 150 / dt / 1e6:.2f} M tokens/s "
          f"({dt / n * 1e3:.3f} ms per batch incl. the host-side compares), last status {model.decoder.group_status()}")

m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0, enc_scale=16.0).items()})
soak(m.to(dev).eval(), "cnn_lstm, 1 encoder stream", int(os.environ.get("N1", "6000")), 1)
enc_p = dict(img_height=64, img_width=320, channels=3, model_name="resnet50", embedding_dim=256, freeze_backbone=True)
r = Seq2SeqModel("resnet_lstm", cfg["vocab_size"], enc_p, synth.decoder_params(cfg))
shapes = [(k, tuple(v.shape)) for k, v in r.encoder.state_dict().items()]
full = {"encoder." + k: torch.from_numpy(v) for k, v in synth.make_resnet_state_dict(shapes, seed=5).items()}
full.update({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0).items() if k.startswith("decoder.")})
r.load_state_dict(full)
soak(r.to(dev).eval(), "resnet50_lstm, 2 encoder streams", int(os.environ.get("N2", "1500")), 2)
