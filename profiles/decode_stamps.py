import ctypes, sys, torch
sys.path.insert(0, "hmer-img2latex_amd")
from img2latex_amd import synth, _lib
_lib.LIB_PATH = "hmer-img2latex_amd/csrc/build/libimg2latex_hip_stamps.so"      # built by profiles/run_stamps.sh
from img2latex_amd.model import Seq2SeqModel
cfg = synth.model_config()
dev = torch.device("cuda:0")
model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg)).to(dev).eval()
enc = torch.randn(256, 256, device=dev)
L = _lib.lib()
L.i2l_debug_group_status_offset.restype = ctypes.c_size_t
off = L.i2l_debug_group_status_offset(256, 512, 256, 256, 1)
for it in range(3):
    ids, _ = model.greedy_ids(enc, 1, 2, 150)
torch.cuda.synchronize()
ws = model.decoder._ws
st = ws[off:off + 2048].view(torch.int32).cpu().tolist()
names = ["A matvec", "B cand poll", "C cell+P", "D h poll", "D2 barrier+h_s", "E logits", "F argmax+publish"]
for blk in (0, 8, 16, 24, 1):
    v = st[8 + blk * 8: 16 + blk * 8]
    tot = sum(v[:7])
    print("wg", blk, "total us", tot / 100.0, " ".join(f"{n}={x/100.0/150:.2f}" for n, x in zip(names, v)))
