"""Evaluation metrics (reference img2latex/training/metrics.py).

CPU: the oracle restatement against the reference's golden outputs (tests/golden/metrics.npz).
GPU: the HIP kernels (through the C ABI / the drop-in functions) against the oracle's integer statistics (exact)
and against the reference's float outputs (exact: the float formulas are the reference's own float64 arithmetic)."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN as GOLDEN_DIR
import metrics_oracle as MO
from img2latex_amd import synth

PAD = 0


def load():
    d = np.load(f"{GOLDEN_DIR}/metrics.npz")
    preds = [d["pred"][i, : d["pred_len"][i]].tolist() for i in range(len(d["pred_len"]))]
    tgts = [d["tgt"][i, : d["tgt_len"][i]].tolist() for i in range(len(d["tgt_len"]))]
    return d, preds, tgts


def ma_inputs(d):
    s1, s2, B, T, V = (int(x) for x in d["ma_seed"])
    logits = np.round(synth.normal_like(s1, "logits", (B, T, V)) * 4.0).astype(np.float32) / 4.0
    targets = synth.randint(s2, "targets", (B, T), 0, V).astype(np.int64)
    return logits, targets


def test_oracle_matches_reference_outputs():
    d, preds, tgts = load()
    for i, (p, t) in enumerate(zip(preds, tgts)):
        assert MO.levenshtein_distance(p, t) == d["lev"][i], i
        for n in (1, 2, 3, 4):
            assert MO.bleu_n_score(p, t, n) == d["bleu"][i, n - 1], (i, n)
    cm = MO.calculate_metrics(preds, tgts)
    assert cm["bleu"] == d["calc"][0] and cm["levenshtein"] == d["calc"][1] and cm["batch_size"] == d["calc"][2]
    assert MO.token_list_accuracy(preds, tgts, PAD) == tuple(d["tla"].tolist())
    logits, targets = ma_inputs(d)
    assert MO.masked_accuracy(logits, targets, PAD) == tuple(d["ma"].tolist())


def test_host_bleu_finalisation_is_bit_identical_to_the_reference_formula():
    """The float64 tail that turns the kernel's integer match counts into BLEU (host logic of the product) against
    the oracle -- itself pinned to the reference's outputs above -- on the golden pairs and 2000 seeded random pairs,
    n = 1..4, incl. empty / shorter-than-n / zero-match sequences: equal to the last bit."""
    import collections
    from img2latex_amd.training.metrics import _bleu_from_counts
    d, preds, tgts = load()
    rng = np.random.RandomState(7)
    for _ in range(2000):
        preds.append(rng.randint(0, 6, size=rng.randint(0, 30)).tolist())
        tgts.append(rng.randint(0, 6, size=rng.randint(0, 30)).tolist())
    for p, t in zip(preds, tgts):
        match = []
        for g in range(1, 5):
            pc = collections.Counter(tuple(p[i:i + g]) for i in range(len(p) - g + 1))
            tc = collections.Counter(tuple(t[i:i + g]) for i in range(len(t) - g + 1))
            match.append(sum(min(c, tc[k]) for k, c in pc.items()) if min(len(p), len(t)) >= g else 0)
        for n in (1, 2, 3, 4):
            assert _bleu_from_counts(match, len(p), len(t), n) == MO.bleu_n_score(p, t, n)


@pytest.mark.gpu
def test_kernel_statistics_and_scores_vs_reference():
    from img2latex_amd.training import metrics as M
    d, preds, tgts = load()
    st = M.sequence_statistics(preds, tgts, 4, PAD)
    for i, (p, t) in enumerate(zip(preds, tgts)):
        assert int(st["lev"][i]) == MO.levenshtein_raw(p, t), i
        assert st["match"][i].tolist() == MO.ngram_matches(p, t, 4), i
    cm = M.calculate_metrics(preds, tgts)
    assert cm["bleu"] == d["calc"][0] and cm["levenshtein"] == d["calc"][1] and cm["batch_size"] == int(d["calc"][2])
    assert M.token_list_accuracy(preds, tgts, PAD) == tuple(d["tla"].tolist())
    for i in (0, 1, 2, 5, 9, 13, 17, 18):                                 # single-pair entry points
        assert M.levenshtein_distance(preds[i], tgts[i]) == d["lev"][i]
        for n in (1, 2, 3, 4):
            assert M.bleu_n_score(preds[i], tgts[i], n) == d["bleu"][i, n - 1]


@pytest.mark.gpu
def test_masked_accuracy_on_device():
    from img2latex_amd.training import metrics as M
    d, _, _ = load()
    logits, targets = ma_inputs(d)
    dev = torch.device("cuda:0")
    got = M.masked_accuracy(torch.from_numpy(logits).to(dev), torch.from_numpy(targets).to(dev), PAD)
    assert got == tuple(d["ma"].tolist())
    # a full-size training batch: (64, 149, 512) logits never leave the device; checked against torch on the device
    g = torch.Generator(device="cpu").manual_seed(3)
    lg = torch.randn(64, 149, 512, generator=g).to(dev)
    tg = torch.randint(0, 512, (64, 149), generator=g).to(dev)
    c, t = M.masked_accuracy(lg, tg, PAD)
    mask = tg.ne(PAD)
    assert t == int(mask.sum()) and c == int((lg.argmax(-1).eq(tg) & mask).sum())


@pytest.mark.gpu
def test_device_id_matrices_and_long_sequences():
    """Ids straight from a decode kernel layout ((B, steps) int32 + lengths), sequences longer than a workgroup."""
    from img2latex_amd.training import metrics as M
    dev = torch.device("cuda:0")
    B, W = 9, 700
    P = synth.randint(31, "p", (B, W), 1, 6).astype(np.int32)
    T = synth.randint(32, "t", (B, W), 1, 6).astype(np.int32)
    pl = np.array([0, 1, 50, 255, 256, 257, 513, 699, 700], np.int32)
    tl = np.array([3, 0, 64, 300, 256, 2, 700, 650, 700], np.int32)
    st = M.device_sequence_statistics(torch.from_numpy(P).to(dev), torch.from_numpy(pl).to(dev),
                                      torch.from_numpy(T).to(dev), torch.from_numpy(tl).to(dev), 4, PAD)
    for i in range(B):
        p, t = P[i, : pl[i]].tolist(), T[i, : tl[i]].tolist()
        assert int(st["lev"][i]) == MO.levenshtein_raw(p, t), i
        assert st["match"][i].tolist() == MO.ngram_matches(p, t, 4), i


@pytest.mark.gpu
def test_very_long_sequences():
    """max_len 1400 (five passes of the workgroup per anti-diagonal, 28 KB of LDS): same integers as the oracle (lengths
    straddling the workgroup size, repeated tokens, one empty side)."""
    from img2latex_amd.training import metrics as M
    dev = torch.device("cuda:0")
    B, W = 4, 1400
    P = synth.randint(41, "p", (B, W), 1, 5).astype(np.int32)
    T = synth.randint(42, "t", (B, W), 1, 5).astype(np.int32)
    pl = np.array([1400, 63, 0, 1301], np.int32)
    tl = np.array([1399, 1400, 7, 65], np.int32)
    st = M.device_sequence_statistics(torch.from_numpy(P).to(dev), torch.from_numpy(pl).to(dev),
                                      torch.from_numpy(T).to(dev), torch.from_numpy(tl).to(dev), 4, PAD)
    for i in range(B):
        p, t = P[i, : pl[i]].tolist(), T[i, : tl[i]].tolist()
        assert int(st["lev"][i]) == MO.levenshtein_raw(p, t), i
        assert st["match"][i].tolist() == MO.ngram_matches(p, t, 4), i


@pytest.mark.gpu
@pytest.mark.parametrize("width", [1, 63, 64, 65, 128, 150, 192, 193, 256])
def test_bit_vector_levenshtein_block_boundaries(width):
    """max_len <= 256 takes the bit-vector kernel (Myers' block algorithm, one 64-token block per word): every block
    count, lengths on and around the block boundaries, tiny and large alphabets (match masks dense / sparse), one side
    empty, identical and disjoint pairs -- the same integers as the oracle's table walk (metrics.py:63-83)."""
    from img2latex_amd.training import metrics as M
    dev = torch.device("cuda:0")
    B = 24
    lens = sorted({0, 1, 2, width // 2, max(width - 1, 0), width, min(63, width), min(64, width), min(65, width),
                   min(127, width), min(129, width)})
    for alpha in (2, 5, 400):
        P = synth.randint(100 + width, f"p{alpha}", (B, width), 1, 1 + alpha).astype(np.int32)
        T = synth.randint(200 + width, f"t{alpha}", (B, width), 1, 1 + alpha).astype(np.int32)
        T[3] = P[3]                                              # identical pair
        T[4] = P[4] + 1000                                       # no token in common
        T[5, : width // 2] = P[5, width - width // 2:]           # shifted overlap
        pl = np.array([lens[i % len(lens)] for i in range(B)], np.int32)
        tl = np.array([lens[(i * 7 + 3) % len(lens)] for i in range(B)], np.int32)
        pl[3] = tl[3] = width
        pl[4] = tl[4] = width
        st = M.device_sequence_statistics(torch.from_numpy(P).to(dev), torch.from_numpy(pl).to(dev),
                                          torch.from_numpy(T).to(dev), torch.from_numpy(tl).to(dev), 4, PAD, _max_len=width)
        for i in range(B):
            p, t = P[i, : pl[i]].tolist(), T[i, : tl[i]].tolist()
            assert int(st["lev"][i]) == MO.levenshtein_raw(p, t), (width, alpha, i, pl[i], tl[i])
            assert st["match"][i].tolist() == MO.ngram_matches(p, t, 4), (width, alpha, i)
        st2 = M.device_sequence_statistics(torch.from_numpy(P).to(dev), torch.from_numpy(pl).to(dev),
                                           torch.from_numpy(T).to(dev), torch.from_numpy(tl).to(dev), 2, PAD, _max_len=width)
        for i in range(B):                                       # max_n below 4: the higher orders stay 0
            p, t = P[i, : pl[i]].tolist(), T[i, : tl[i]].tolist()
            assert st2["match"][i].tolist()[:2] == MO.ngram_matches(p, t, 2), (width, alpha, i)
            assert st2["match"][i].tolist()[2:] in ([], [0, 0]), (width, alpha, i)


def test_scores_host_helper_is_bit_identical_to_the_python_formulas():
    """i2l_scores_from_statistics (r04: the float64 tail of calculate_metrics in the library's host code, so that the evaluate
    chain does not spend 0.5 ms per batch in the interpreter) == metrics_from_statistics (the reference's formulas, metrics.py:
    85-94,113-181, in Python) on random statistics with the edge cases: empty prediction / target, zero n-gram hits,
    sequences shorter than n, brevity penalty on and off; padded rows (stride 10, as Predictor's landing buffer)."""
    import torch
    from img2latex_amd.training import metrics as M
    rng = np.random.default_rng(5)
    P = 600
    gl = rng.integers(0, 160, P)
    tl = rng.integers(0, 160, P)
    gl[:8] = [0, 0, 5, 3, 2, 1, 150, 150]
    tl[:8] = [0, 7, 0, 3, 9, 1, 150, 149]
    match = np.stack([rng.integers(0, np.maximum(gl - g, 0) + 1) for g in range(4)], axis=1)
    match[rng.random(P) < 0.2, 3] = 0
    lev = rng.integers(0, np.maximum(np.maximum(gl, tl), 1) + 1)
    packed = np.zeros((P, 10), np.int32)
    packed[:, 0], packed[:, 1:5], packed[:, 7], packed[:, 8], packed[:, 9] = lev, match, gl, tl, 12345
    host = torch.from_numpy(packed)
    want = M.metrics_from_statistics(M.unpack_statistics(host[:, :9]))
    got = M.metrics_from_packed(host)
    assert got == want, (got, want)                                   # float64 equality, not closeness
    one = M.metrics_from_packed(host[6:7])
    assert one == M.metrics_from_statistics(M.unpack_statistics(host[6:7, :9]))
    with pytest.raises(TypeError):
        M.metrics_from_packed(host[:, :5])
