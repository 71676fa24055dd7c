"""Do the decode groups keep their one-XCD placement inside the co-resident pipeline?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "hmer-img2latex_amd"))
from img2latex_amd import _lib, synth
from img2latex_amd.model import Seq2SeqModel
from img2latex_amd.pipeline import GreedyPipeline
dev = torch.device("cuda:0")
cfg = synth.model_config()
model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0, enc_scale=16.0).items()})
model = model.to(dev).eval()
x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
for stag in (None, False):
    pipe = GreedyPipeline(model, synth.START, synth.END, 150, rows_per_workgroup=0, decode_flags=_lib.FLAG_DECODE_GROUP8, decode_priority=-1, hold_encoder=stag)
    seen = []
    for i in range(40):
        if pipe.pending() >= pipe.depth:
            pipe.collect()
            seen.append(model.decoder.group_status())      # status of the most recently PREPARED slot: the decode in flight or just done
        pipe.submit(x)
    while pipe.pending():
        pipe.collect()
    torch.cuda.synchronize()
    print("hold_encoder", stag, "last:", model.decoder.group_status(), "one-XCD counts seen:", sorted(set(s["groups_on_one_xcd"] for s in seen if s)))
