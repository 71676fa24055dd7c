"""Per-phase times of decode_group16_kernel INSIDE the co-resident pipeline (beside the next batch's conv workgroups)
against the same kernel alone: which phases stretch?  (stamps build: bash profiles/run_stamps.sh build)"""
import ctypes, sys, torch
sys.path.insert(0, "hmer-img2latex_amd")
from img2latex_amd import synth, _lib
_lib.LIB_PATH = "hmer-img2latex_amd/csrc/build/libimg2latex_hip_stamps.so"
from img2latex_amd.model import Seq2SeqModel
from img2latex_amd.pipeline import GreedyPipeline
cfg = synth.model_config()
dev = torch.device("cuda:0")
model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0, enc_scale=16.0).items()})
model = model.to(dev).eval()
x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
L = _lib.lib()
L.i2l_debug_group_status_offset.restype = ctypes.c_size_t
off = L.i2l_debug_group_status_offset(256, 512, 256, 256, 1)
names = ["A", "B candwait", "C cell", "D hpoll", "barrier", "E logits", "F argmax"]

def show(ws, tag):
    st = ws[off:off + 2048].view(torch.int32).cpu().tolist()
    for blk in (0, 9):
        for half, role in ((0, "w0"), (1, "w2")):
            v = st[8 + blk * 16 + half * 8: 16 + blk * 16 + half * 8]
            print(f"{tag} wg{blk} {role}: {sum(v[:7]) / 100.0 / 150:.2f} us/step  " + " ".join(f"{n}={x_/100.0/150:.2f}" for n, x_ in zip(names, v)))

with torch.no_grad():
    enc = model.encoder(x)
    for _ in range(3):
        model.greedy_ids(enc, synth.START, synth.END, 150, flags=_lib.FLAG_DECODE_GROUP16)
    torch.cuda.synchronize()
    show(model.decoder._ws, "alone ")
pipe = GreedyPipeline(model, synth.START, synth.END, 150, rows_per_workgroup=0, decode_flags=_lib.FLAG_DECODE_GROUP16, decode_priority=-1)
for i in range(60):
    if pipe.pending() >= pipe.depth:
        pipe.collect()
    pipe.submit(x)
while pipe.pending():
    pipe.collect()
torch.cuda.synchronize()
slots = [v[0] for k, v in model.decoder._ws_by_stream.items() if isinstance(k, tuple) and k[0] == "slot"]
show(slots[0], "beside")          # a slot whose last decode ran beside an encoder (not the final, un-overlapped one)
show(slots[-1], "beside")
