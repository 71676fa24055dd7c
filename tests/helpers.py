"""Shared helpers for the parity tests: fixtures -> (cfg, state_dict, inputs)."""
import json
import os

import numpy as np
import torch

from img2latex_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
START, END, PAD = synth.START, synth.END, synth.PAD
SMALL = ["tiny_l1", "tiny_l2_attn", "odd_dims"]
ALL = SMALL + ["primary", "secondary"]


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
