"""Per-phase times of decode_group16_kernel (in-kernel wall clock; build: bash profiles/run_stamps.sh build)."""
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
    ids, _ = model.greedy_ids(enc, 1, 2, 150, flags=_lib.FLAG_DECODE_GROUP16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ids, _ = model.greedy_ids(enc, 1, 2, 150, flags=_lib.FLAG_DECODE_GROUP16); e1.record(); torch.cuda.synchronize()
print(f"prepare + decode16: {e0.elapsed_time(e1):.3f} ms")
ws = model.decoder._ws
st = ws[off:off + 2048].view(torch.int32).cpu().tolist()
names = ["A gates mfma", "B cand wait", "C cell+publish", "D h poll", "D2 barrier", "E logits mfma+handoff", "F argmax+publish"]
for blk in (0, 1, 8, 15):
    for half, role in ((0, "wave 0 (argmax)"), (1, "wave 2 (partial)")):
        v = st[8 + blk * 16 + half * 8: 16 + blk * 16 + half * 8]
        print("wg", blk, role, "total us/step", round(sum(v[:7]) / 100.0 / 150, 2), " ".join(f"{n}={x/100.0/150:.2f}" for n, x in zip(names, v)))
