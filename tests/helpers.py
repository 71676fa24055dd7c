"""Shared helpers for the parity tests: fixtures -> (cfg, state_dict, inputs)."""
import json
import os

import numpy as np
import torch

from img2latex_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
START, END, PAD = synth.START, synth.END, synth.PAD
SMALL = ["tiny_l1", "tiny_l2_attn", "odd_dims"]
WIDE = ["ref_test_64x800", "shipped_128x800"]      # the reference's own shapes (tests/test_encoder.py:11-42, configs/config.yaml:30-50)
BIG = ["primary", "secondary"] + WIDE               # fixtures that hold samples instead of whole tensors, T = 24
ALL = SMALL + BIG


def load(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = json.loads(str(d["cfg_json"]))
    sd_kw = json.loads(str(d["sd_kw_json"]))
    if "end_clock" in sd_kw and sd_kw["end_clock"] is not None:
        sd_kw["end_clock"] = tuple(sd_kw["end_clock"])
    return d, cfg, sd_kw


_SD_CACHE = {}


def np_state_dict(name):
    if name not in _SD_CACHE:
        _, cfg, sd_kw = load(name)
        _SD_CACHE[name] = synth.make_state_dict(cfg, **sd_kw)
    return _SD_CACHE[name]


def torch_state_dict(name, device="cpu"):
    return {k: torch.from_numpy(v.copy()).to(device) for k, v in np_state_dict(name).items()}


def images(cfg, batch=4, seed=1234, device="cpu"):
    return torch.from_numpy(synth.make_images(batch, cfg, seed=seed)).to(device)


def padded_to_lists(arr, lens):
    return [list(map(int, arr[j, :lens[j]])) for j in range(len(lens))]


def sample(t, n=4096):
    f = t.detach().reshape(-1).cpu()
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy()


# ---------------------------------------------------------------------------------------------------------------
# Discrete decisions of the conv blocks (pooling arg max, ReLU gate).  The conv gradients are discontinuous in them,
# and a near-tie (two window values, or a pre-activation and zero, closer than fp32 rounding resolves) is decided
# differently by different correct fp32 evaluations -- ATen's included.  The training tests therefore (1) read the
# choices the HIP forward made, (2) require that they differ from a float64 evaluation only at such near-ties, and
# (3) compare the HIP gradients element by element with the float64 gradient of the SAME choices
# (oracle.conv_block_decided), where the remaining difference is smooth rounding only.
# ---------------------------------------------------------------------------------------------------------------
def hip_decisions(model, x_dev):
    """Per conv block (argmax map, ReLU gate) of the HIP training forward (deterministic: same maps as inside a step)."""
    am = []
    with torch.no_grad():
        ys = model.encoder.conv_blocks(x_dev, am)
    return [(a.cpu().long(), (y > 0).cpu()) for a, y in zip(am, ys)]


def check_decisions(sd, cfg, x, decisions, tol=2e-6):
    """Against float64 (each block evaluated on the float64 output of the blocks before it UNDER the given decisions):
    a decision may differ only where the gap is below `tol` * sum |x||w| of the window -- what an fp32 dot product
    of that length cannot resolve.  Returns (number of windows deciding differently, number of windows)."""
    import torch.nn.functional as F
    import img2latex_oracle as O
    inp = x.double()
    n_off = n_all = 0
    for i, (am, gate) in enumerate(decisions):
        w = sd[f"encoder.cnn_layers.{3 * i}.weight"].double()
        b = sd[f"encoder.cnn_layers.{3 * i}.bias"].double()
        win = O.pool_windows(F.conv2d(inp, w, None, padding=1))
        mag = O.pool_windows(F.conv2d(inp.abs(), w.abs(), None, padding=1)).amax(-1) + b.abs()[None, :, None, None]
        top2 = win.topk(2, dim=-1).values
        gap = top2[..., 0] - top2[..., 1]
        pre = top2[..., 0] + b[None, :, None, None]
        wrong_gate = gate != (pre > 0)
        wrong_am = (am != win.argmax(-1)) & gate & (pre > 0) & (gap > 0)       # the arg max matters where the ReLU passes
        if wrong_am.any():
            # the chosen value, not just the runner-up, must be within tol of the maximum
            chosen = win.gather(-1, am.unsqueeze(-1)).squeeze(-1)
            worst = float(((top2[..., 0] - chosen)[wrong_am] / mag[wrong_am]).max())
            assert worst <= tol, f"block {i}: arg max differs from float64 at a gap of {worst:.2e} of sum|x||w|"
        if wrong_gate.any():
            worst = float((pre[wrong_gate].abs() / mag[wrong_gate]).max())
            assert worst <= tol, f"block {i}: ReLU gate differs from float64 at |pre| = {worst:.2e} of sum|x||w|"
        tie = (gap == 0) & gate & (pre > 0)
        assert bool((am[tie] == win.argmax(-1)[tie]).all()), f"block {i}: an exact tie must take the first index"
        n_off += int(wrong_am.sum()) + int(wrong_gate.sum())
        n_all += am.numel()
        inp = O.conv_block_decided(inp, w, b, am, gate)
    return n_off, n_all


def adam_first_step_allowance(g_a, g_b, p0, coef_a=1.0, coef_b=1.0, coef_unc=0.0, lr=1e-3, wd=1e-4, eps=1e-8, base=3e-6):
    """How far one parameter may move between two runs of the FIRST Adam step whose gradients are g_a and g_b and whose
    clip coefficients are coef_a and coef_b (each known to a relative `coef_unc`: the total norm is a sum over 11.6 M
    terms that torch accumulates in fp32 and the HIP kernel in double):
        p -= lr * f(x),  x = coef * g + wd * p0,  f(x) = x / (|x| + eps),  |f(a) - f(b)| <= min(2, |a - b| / (min(|a|, |b|) + eps)).
    An element whose |x| is at the level of x's own rounding error -- a tiny gradient, or a clipped gradient that happens
    to cancel the weight-decay term -- moves by up to lr either way."""
    xa, xb = coef_a * g_a + wd * p0, coef_b * g_b + wd * p0
    slack = coef_unc * torch.maximum((coef_a * g_a).abs(), (coef_b * g_b).abs())
    dx = (xa - xb).abs() + slack
    lo = torch.clamp(torch.minimum(xa.abs(), xb.abs()) - slack, min=0.0)
    return base + lr * torch.clamp(dx / (lo + eps), max=2.0)
