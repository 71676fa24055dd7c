import sys, os, torch
sys.path.insert(0, "hmer-img2latex_amd")
from img2latex_amd import _lib, synth
from img2latex_amd.model import Seq2SeqModel
dev = torch.device("cuda:0")
cfg = synth.model_config()
m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0, enc_scale=16.0).items()})
m = m.to(dev).eval()
x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
with torch.no_grad():
    enc = m.encoder(x)
    for fl, nm in ((0, "group4"), (_lib.FLAG_DECODE_GROUP8, "group8")):
        for _ in range(3): m.greedy_ids(enc, 1, 2, 150, flags=fl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ids, _ = m.greedy_ids(enc, 1, 2, 150, flags=fl)
        e1.record(); torch.cuda.synchronize()
        print(nm, e0.elapsed_time(e1) / 20, "ms per prepare+decode", m.decoder.group_status(), int(ids.min()))
