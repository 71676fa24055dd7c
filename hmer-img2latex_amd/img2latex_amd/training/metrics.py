"""Evaluation metrics with the reference's names and results (img2latex/training/metrics.py), computed on the
device: the id sequences and the (B,T,V) logits stay in HBM, the kernels return the INTEGER statistics
(Levenshtein table corner, clipped n-gram matches, accuracy counts) and the few float64 operations per pair
that turn them into scores are written here exactly as the reference writes them.

``calculate_metrics`` / ``masked_accuracy`` / ``token_list_accuracy`` are drop-ins for the calls at
cli.py:495, trainer.py:391,526 and metrics.py:591-621.  No CPU fallback: inputs are uploaded if they are lists.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple, Union

import torch

from .. import _lib

Seqs = Union[Sequence[Sequence[int]], torch.Tensor]


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("img2latex_amd: the metrics kernels need the ROCm device; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _pack(seqs: Sequence[Sequence[int]], dev: torch.device, width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Ragged id lists -> (padded (n, width) int32, lengths (n) int32) on the device."""
    import numpy as np
    n = len(seqs)
    ids = np.zeros((n, max(width, 1)), dtype=np.int32)
    lens = np.empty((n,), dtype=np.int32)
    for i, s in enumerate(seqs):
        if isinstance(s, torch.Tensor):
            s = s.tolist()
        lens[i] = len(s)
        if len(s):
            ids[i, : len(s)] = s
    return torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev)


def sequence_statistics(predictions: Sequence[Sequence[int]], targets: Sequence[Sequence[int]], n: int = 4,
                        pad_token_id: int = 0) -> Dict[str, torch.Tensor]:
    """One launch for the whole batch.  Returns host int tensors: lev (P), match (P,4), tla (P,2), gen_len, true_len."""
    assert len(predictions) == len(targets), "Predictions and targets must have the same length"
    pairs = len(predictions)
    if pairs == 0:
        z = torch.zeros((0,), dtype=torch.int32)
        return dict(lev=z, match=torch.zeros((0, 4), dtype=torch.int32), tla=torch.zeros((0, 2), dtype=torch.int32),
                    gen_len=z, true_len=z)
    dev = _device()
    width = max([len(s) for s in predictions] + [len(s) for s in targets] + [1])
    p_ids, p_len = _pack(predictions, dev, width)
    t_ids, t_len = _pack(targets, dev, width)
    return device_sequence_statistics(p_ids, p_len, t_ids, t_len, n, pad_token_id, _max_len=width)


def device_sequence_statistics(p_ids: torch.Tensor, p_len: torch.Tensor, t_ids: torch.Tensor, t_len: torch.Tensor,
                               n: int = 4, pad_token_id: int = 0, _max_len: int = None, defer: bool = False):
    """Same for id matrices that already live on the device ((P, W) int32 + lengths), e.g. the decode kernel's output.
    ``defer=True`` returns the statistics as ONE device tensor (P, 9) int32 = [lev | match x4 | tla x2 | gen_len |
    true_len] without touching the host (``unpack_statistics`` splits its host copy), so a caller can overlap the next
    batch's host work with this one's kernels."""
    for name, t in (("pred ids", p_ids), ("pred lengths", p_len), ("target ids", t_ids), ("target lengths", t_len)):
        _lib.require_gpu(t, name, torch.int32)
    pairs = p_ids.shape[0]
    if _max_len is None:                                             # one sync; the list entry point knows it already
        pm, tm = int(p_len.max()), int(t_len.max())
        if pm > p_ids.shape[1] or tm > t_ids.shape[1]:
            raise RuntimeError("sequence length exceeds the id matrix width")
        max_len = max(pm, tm, 0)
    else:
        max_len = int(_max_len)
    if p_ids.shape[1] < max_len or t_ids.shape[1] < max_len:        # the kernel indexes rows up to max_len
        pad = lambda m: torch.nn.functional.pad(m, (0, max_len - m.shape[1]))
        p_ids, t_ids = pad(p_ids).contiguous(), pad(t_ids).contiguous()
    dev = p_ids.device
    lev = torch.empty((pairs,), dtype=torch.int32, device=dev)
    match = torch.empty((pairs, 4), dtype=torch.int32, device=dev)
    tla = torch.empty((pairs, 2), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().i2l_sequence_metrics(
        p_ids.data_ptr(), p_len.data_ptr(), p_ids.stride(0), t_ids.data_ptr(), t_len.data_ptr(), t_ids.stride(0), pairs,
        max_len, int(n), int(pad_token_id), lev.data_ptr(), match.data_ptr(), tla.data_ptr(), _lib.stream_ptr()),
        "sequence_metrics")
    packed = torch.cat([lev[:, None], match, tla, p_len[:, None], t_len[:, None]], dim=1)
    return packed if defer else unpack_statistics(packed.cpu())


def unpack_statistics(packed_host: torch.Tensor) -> Dict[str, torch.Tensor]:
    """(P, 9) int32 host tensor of device_sequence_statistics(defer=True) -> the dict of the other entry points."""
    return dict(lev=packed_host[:, 0], match=packed_host[:, 1:5], tla=packed_host[:, 5:7], gen_len=packed_host[:, 7],
                true_len=packed_host[:, 8])


def _lev_similarity(raw_distance: int, rows: int, cols: int) -> float:
    """metrics.py:85-94."""
    max_length = max(rows, cols)
    if max_length == 0:
        return 1.0
    return 1.0 - (raw_distance / max_length)


def _bleu_from_counts(match: Sequence[int], gen_len: int, true_len: int, n: int) -> float:
    """BLEU-n from the kernel's clipped n-gram match counts (i2l_sequence_metrics).  Bit-identical to the
    reference's score (metrics.py:113-181) because the float64 operations run in the same order: each
    precision is ONE division match / (gen_len - g + 1); their logs are added left to right starting from
    0.0; one division by n; one exp; the brevity factor exp(1 - true_len / gen_len) multiplies from the left."""
    if min(gen_len, true_len) == 0:
        return 0.0
    log_sum = 0.0
    for g in range(1, n + 1):
        hits = int(match[g - 1]) if min(gen_len, true_len) >= g else 0
        if hits == 0:                       # a zero precision zeroes the geometric mean (also: too short for g-grams)
            return 0.0
        log_sum += math.log(hits / (gen_len - g + 1))
    score = math.exp(log_sum / n)
    if gen_len >= true_len:
        return score
    return math.exp(1.0 - true_len / gen_len) * score


def levenshtein_distance(sequence_one: List[int], sequence_two: List[int]) -> float:
    """Normalised Levenshtein similarity of one pair (metrics.py:49-94)."""
    st = sequence_statistics([sequence_one], [sequence_two])
    return _lev_similarity(int(st["lev"][0]), len(sequence_one), len(sequence_two))


def bleu_n_score(generated_sequence: List[int], true_sequence: List[int], n: int = None) -> float:
    """BLEU-n of one pair (metrics.py:97-181)."""
    n = 4 if n is None else n
    if n > 4:
        raise NotImplementedError("img2latex_amd: BLEU-n is built for n <= 4 (the reference calls it with 4)")
    st = sequence_statistics([generated_sequence], [true_sequence], n)
    return _bleu_from_counts(st["match"][0].tolist(), len(generated_sequence), len(true_sequence), n)


def calculate_metrics(predictions: List[List[int]], targets: List[List[int]]) -> Dict[str, float]:
    """metrics.py:184-223: mean BLEU-4 and mean Levenshtein similarity of a batch, one kernel launch."""
    assert len(predictions) == len(targets), "Predictions and targets must have the same length"
    num_sequences = len(predictions)
    return metrics_from_statistics(sequence_statistics(predictions, targets, 4))


def metrics_from_packed(packed_host: torch.Tensor) -> Dict[str, float]:
    """The float64 tail of calculate_metrics on the (P, >= 9) int32 host rows of device_sequence_statistics(defer=True), by
    the library's host helper (i2l_scores_from_statistics: the formulas below, operation by operation, without the
    interpreter: 0.5 ms -> 5 us per 256 pairs; ``metrics_from_statistics`` is the Python statement of the same thing)."""
    import ctypes
    if packed_host.dtype != torch.int32 or packed_host.dim() != 2 or packed_host.shape[1] < 9 or packed_host.stride(1) != 1:
        raise TypeError("packed statistics must be a (P, >= 9) int32 host tensor with contiguous rows")
    b, l = ctypes.c_double(), ctypes.c_double()
    _lib.check(_lib.lib().i2l_scores_from_statistics(packed_host.data_ptr(), packed_host.shape[0], packed_host.stride(0), 4,
                                                     ctypes.byref(b), ctypes.byref(l)), "scores_from_statistics")
    return {"bleu": b.value, "levenshtein": l.value, "batch_size": int(packed_host.shape[0])}


def metrics_from_statistics(st: Dict[str, torch.Tensor]) -> Dict[str, float]:
    """The float64 tail of calculate_metrics on the kernel's integer statistics (host tensors)."""
    match, lev = st["match"].tolist(), st["lev"].tolist()
    gl, tl = st["gen_len"].tolist(), st["true_len"].tolist()
    num_sequences = len(lev)
    bleu_scores = [_bleu_from_counts(match[i], gl[i], tl[i], 4) for i in range(num_sequences)]
    mean_bleu = sum(bleu_scores) / num_sequences
    lev_similarities = [_lev_similarity(lev[i], gl[i], tl[i]) for i in range(num_sequences)]
    mean_lev = sum(lev_similarities) / num_sequences
    return {"bleu": mean_bleu, "levenshtein": mean_lev, "batch_size": num_sequences}


def token_list_accuracy(predictions: List[List[int]], targets: List[List[int]], pad_token_id: int) -> Tuple[int, int]:
    """metrics.py:241-277: (correct, non-pad) token counts over the common prefix length of every pair."""
    if len(predictions) == 0:
        return 0, 0
    k = min(len(predictions), len(targets))                     # zip() semantics
    st = sequence_statistics(predictions[:k], targets[:k], 1, pad_token_id)
    tla = st["tla"].to(torch.int64).sum(dim=0)
    return int(tla[0]), int(tla[1])


def masked_accuracy(logits: torch.Tensor, targets: torch.Tensor, pad_token_id: int) -> Tuple[int, int]:
    """metrics.py:226-238 without moving the (B,T,V) logits to the host: (correct, total) over non-pad targets."""
    logits = _lib.require_gpu(logits.detach(), "logits").contiguous()
    targets = targets.detach()
    if not targets.is_cuda:
        targets = targets.to(logits.device)
    targets = targets.to(torch.int64).contiguous()
    V = logits.shape[-1]
    rows = logits.numel() // V
    if targets.numel() != rows:
        raise RuntimeError(f"targets must have {rows} elements, got {targets.numel()}")
    out = torch.empty((2,), dtype=torch.int64, device=logits.device)
    _lib.check(_lib.lib().i2l_masked_accuracy(logits.data_ptr(), targets.data_ptr(), rows, V, int(pad_token_id),
                                              out.data_ptr(), _lib.stream_ptr()), "masked_accuracy")
    c, t = out.cpu().tolist()
    return int(c), int(t)
