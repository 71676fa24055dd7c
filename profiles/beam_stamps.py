"""Phase times of beam_group_kernel (library built with EXTRA=-DI2L_GROUP_STAMPS): per-step microseconds of
A (recurrent matvec), B1 (candidate merge), B2 (selection + bookkeeping), C (cell), D-poll, D-write, E (logits), F (local top-k)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hmer-img2latex_amd"))
from img2latex_amd import _lib, synth  # noqa: E402
from img2latex_amd.model import Seq2SeqModel  # noqa: E402


def main():
    n, k, T = 128, 5, 150
    dev = torch.device("cuda:0")
    cfg = synth.model_config(attention=True)
    sd_kw = dict(seed=42, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))
    model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
    model.load_state_dict({k_: torch.from_numpy(v) for k_, v in synth.make_state_dict(cfg, **sd_kw).items()})
    model = model.to(dev).eval()
    images = torch.from_numpy(synth.make_images(n, cfg, seed=1234)).to(dev)
    L = _lib.lib()
    with torch.no_grad():
        enc = model.encoder(images)
        dec = model.decoder
        w, keep, enc = dec.prepare(enc)
        nbytes = L.i2l_beam_workspace_bytes(n, k, dec.hidden_dim, dec.lstm_layers, T)
        bws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        seq = torch.empty((n, T + 1), dtype=torch.int32, device=dev)
        ln = torch.empty((n,), dtype=torch.int32, device=dev)
        score = torch.empty((n,), dtype=torch.float64, device=dev)
        for _ in range(3):
            _lib.check(L.i2l_beam_decode(ctypes.byref(w), dec._ws.data_ptr(), n, k, T, synth.START, synth.END, bws.data_ptr(), nbytes,
                                         seq.data_ptr(), ln.data_ptr(), score.data_ptr(), 0, _lib.stream_ptr()), "beam")
        torch.cuda.synchronize()
    hist = (2 * n * T * k * 4 + 255) // 256 * 256
    st = bws[hist:hist + 2048].cpu().numpy().view("uint32")
    steps = int(st[4])
    names = ["A", "B1", "B2", "C", "Dpoll", "Dwrite", "E", "F"]
    print("status", st[0], "steps of group 0:", steps, "mean len", float(ln.float().mean()))
    for b in (0, 8, 16, 24, 1):
        v = st[8 + b * 8: 16 + b * 8].astype("float64") / 100.0     # 100 MHz ticks -> us
        print(f"wg {b:2d}: " + "  ".join(f"{nm}={x / max(steps, 1):.2f}" for nm, x in zip(names, v)) + f"  sum={v.sum() / max(steps, 1):.2f} us/step")


if __name__ == "__main__":
    main()
