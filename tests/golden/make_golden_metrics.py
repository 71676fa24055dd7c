#!/usr/bin/env python3
"""Golden vectors for the evaluation metrics: runs the REAL reference functions of
img2latex/training/metrics.py (imported unmodified from /root/reference; same inert torchvision shim as
make_golden.py because img2latex/__init__ pulls the model package) on generated id sequences and stores
inputs + outputs in tests/golden/metrics.npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_metrics.py
"""
import os
import sys
import types

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(REPO, "hmer-img2latex_amd"))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
_tv = types.ModuleType("torchvision")
_tv.__path__ = []
for _sub in ("models", "transforms", "transforms.functional"):
    _m = types.ModuleType("torchvision." + _sub)
    _m.__path__ = []
    sys.modules["torchvision." + _sub] = _m
    setattr(sys.modules["torchvision." + _sub.rsplit(".", 1)[0]] if "." in _sub else _tv, _sub.rsplit(".", 1)[-1], _m)
sys.modules["torchvision"] = _tv

import logging  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from img2latex_amd import synth  # noqa: E402

logging.disable(logging.CRITICAL)
from img2latex.training import metrics as ref  # noqa: E402  (the reference)

PAD = 0


def make_pairs():
    """Ragged pairs that hit: empty sides, equal sequences, prefixes, repeated n-grams (small alphabet), length > 256."""
    pairs = []
    lens = [(0, 0), (0, 5), (4, 0), (1, 1), (3, 3), (7, 12), (12, 7), (30, 30), (40, 25), (150, 150), (149, 131), (300, 280)]
    for k, (lp, lt) in enumerate(lens):
        alpha = 4 if k % 2 == 0 else 23
        p = synth.randint(100 + k, "pred", (lp,), 0, alpha).tolist()
        t = synth.randint(200 + k, "tgt", (lt,), 0, alpha).tolist()
        pairs.append((p, t))
    base = synth.randint(7, "base", (60,), 1, 9).tolist()
    pairs.append((base, list(base)))                       # identical
    pairs.append((base[:20], list(base)))                  # prefix (brevity penalty)
    pairs.append((list(base), base[:20]))
    mut = list(base)
    for i in (3, 17, 18, 44):
        mut[i] = 9
    pairs.append((mut, list(base)))                        # a few substitutions
    pairs.append((base[:10] + base[12:], list(base)))      # deletion
    pairs.append(([5] * 12, [5] * 7))                      # one repeated token: clipping of n-gram counts
    pairs.append(([1, 2, 3, 1, 2, 3, 1, 2], [1, 2, 3, 4, 1, 2]))
    return pairs


def main():
    pairs = make_pairs()
    preds = [p for p, _ in pairs]
    tgts = [t for _, t in pairs]
    width = max(max(len(p) for p in preds), max(len(t) for t in tgts))
    P = np.zeros((len(pairs), width), np.int32)
    T = np.zeros((len(pairs), width), np.int32)
    for i, (p, t) in enumerate(pairs):
        P[i, :len(p)] = p
        T[i, :len(t)] = t
    out = dict(pred=P, tgt=T, pred_len=np.array([len(p) for p in preds], np.int32),
               tgt_len=np.array([len(t) for t in tgts], np.int32))
    out["lev"] = np.array([ref.levenshtein_distance(p, t) for p, t in pairs], np.float64)
    out["bleu"] = np.array([[ref.bleu_n_score(p, t, n) for n in (1, 2, 3, 4)] for p, t in pairs], np.float64)
    cm = ref.calculate_metrics(preds, tgts)
    out["calc"] = np.array([cm["bleu"], cm["levenshtein"], cm["batch_size"]], np.float64)
    out["tla"] = np.array(ref.token_list_accuracy(preds, tgts, PAD), np.int64)
    # masked_accuracy on generated logits / targets (ties included: quantised logits)
    B, TT, V = 6, 37, 29
    logits = np.round(synth.normal_like(11, "logits", (B, TT, V)) * 4.0).astype(np.float32) / 4.0
    targets = synth.randint(12, "targets", (B, TT), 0, V).astype(np.int64)
    out["ma_seed"] = np.array([11, 12, B, TT, V], np.int64)
    out["ma"] = np.array(ref.masked_accuracy(torch.from_numpy(logits), torch.from_numpy(targets), PAD), np.int64)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "metrics.npz"), **out)
    print("pairs", len(pairs), "calc", cm, "tla", out["tla"], "ma", out["ma"])


if __name__ == "__main__":
    main()
