"""How much does each encoder kernel slow the co-resident decode?  decode_group8_kernel (B=256, 150 steps) on one stream while
another stream loops ONE encoder stage (conv block 1 / 2 / 3 / FC) back to back; decode time alone for reference."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "hmer-img2latex_amd"))
from img2latex_amd import _lib, synth
from img2latex_amd.model import Seq2SeqModel
dev = torch.device("cuda:0")
cfg = synth.model_config()
model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0, enc_scale=16.0).items()})
model = model.to(dev).eval()
x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(dev)
enc_m = model.encoder
with torch.no_grad():
    blocks = []
    enc_out = enc_m._forward_impl(x, blocks_out=blocks) if "blocks_out" in enc_m._forward_impl.__code__.co_varnames else enc_m(x)
    enc = enc_m(x)
inputs = [x] + blocks[:2] if blocks else None
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

def stage_fn(i):
    """returns a callable that runs encoder stage i (0..2 conv block, 3 FC) once on the current stream"""
    L = _lib.lib()
    if i < 3:
        conv = enc_m.cnn_layers[3 * i]
        xin = inputs[i]
        B, cin, h, w = xin.shape
        y = torch.empty((B, conv.out_channels, h // 2, w // 2), device=dev)
        nb = L.i2l_conv_workspace_bytes(cin, conv.out_channels)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        def f():
            _lib.check(L.i2l_conv3x3_relu_pool2_fwd(xin.data_ptr(), conv.weight.data_ptr(), conv.bias.data_ptr(), y.data_ptr(), None,
                                                    B, cin, h, w, conv.out_channels, ws.data_ptr(), nb, 0, _lib.stream_ptr()), "conv")
        return f
    feat = blocks[2].reshape(256, -1)
    lin = enc_m.embedding_layer
    out = torch.empty((256, lin.out_features), device=dev)
    nb = L.i2l_linear_workspace_bytes(256, feat.shape[1], lin.out_features)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    def f():
        _lib.check(L.i2l_linear_bias_act_fwd(feat.data_ptr(), lin.weight.data_ptr(), lin.bias.data_ptr(), out.data_ptr(), 256, feat.shape[1],
                                             lin.out_features, 1, ws.data_ptr(), nb, 0, _lib.stream_ptr()), "fc")
    return f

def decode_ms(beside=None, reps=6):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.no_grad():
            with torch.cuda.stream(sa):
                w_, k_, e_ = model.decoder.prepare(enc, slot="t")
                prepared = (w_, k_, e_, model.decoder._ws)
                e0.record(sa)
                model.greedy_ids(enc, synth.START, synth.END, 150, flags=_lib.FLAG_DECODE_GROUP8, prepared=prepared)
                e1.record(sa)
            if beside is not None:
                with torch.cuda.stream(sb):
                    sb.wait_event(e0)           # (r03 added a 30 us delay kernel here; the pipeline now waits for the residency signal)
                    for _ in range(beside[1]):
                        beside[0]()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]

if inputs is None:
    print("encoder does not expose its blocks"); sys.exit(0)
alone = decode_ms()
print(f"decode alone: {alone:.3f} ms")
for i, (name, n) in enumerate((("conv block 1 (3->32)", 14), ("conv block 2 (32->64)", 6), ("conv block 3 (64->128)", 6), ("FC 20480->256", 24))):
    f = stage_fn(i)
    t = decode_ms((f, n))
    print(f"decode beside a loop of {name} x{n}: {t:.3f} ms  (x{t / alone:.2f})")
