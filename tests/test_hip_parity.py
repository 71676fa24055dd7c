"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures generated from the real reference.  Tolerances: token ids exact; logits 1e-4 ABSOLUTE
(BASELINE.json north_star: "logits within 1e-4"); encoder / state 1e-5 relative to max(1,|ref|).
Every measured maximum error is recorded (conftest.record -> gpurun_out/parity_errors.json)."""
import numpy as np
import pytest
import torch

import img2latex_oracle as O
from conftest import record
from helpers import ALL, BIG, END, SMALL, START, images, load, padded_to_lists, sample, torch_state_dict
from img2latex_amd import _lib, synth
from img2latex_amd.model import Seq2SeqModel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, tol, what=None, absolute=False):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = 1.0 if absolute else max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    if what:
        record(what + (" [abs]" if absolute else " [rel to max(1,|ref|)]"), err / scale)
    assert err <= tol * scale, f"{what}: max err {err} > {tol} * {scale}"


_MODELS = {}


def model_for(name, sd_kw=None, cfg=None):
    key = (name, repr(sd_kw))
    if key not in _MODELS:
        if cfg is None:
            _, cfg, kw = load(name)
            sd_kw = kw if sd_kw is None else sd_kw
        m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, **sd_kw).items()})
        _MODELS[key] = (m.to(DEV).eval(), cfg)
    return _MODELS[key]


@pytest.mark.parametrize("name", ALL)
def test_encoder_vs_golden(name):
    d, cfg, _ = load(name)
    m, _ = model_for(name)
    x = images(cfg, device=DEV)
    with torch.no_grad():
        blocks = m.encoder.conv_blocks(x)
        enc = m.encoder(x)
    for i, b in enumerate(blocks):
        assert list(b.shape) == list(d[f"g1_block{i}_shape"])
        if f"g1_block{i}" in d:
            close(b.cpu().numpy(), d[f"g1_block{i}"], 1e-5, f"{name} encoder block{i}")
        else:
            close(sample(b), d[f"g1_block{i}_sample"], 1e-5, f"{name} encoder block{i}")
    close(enc.cpu().numpy(), d["g1_enc"], 1e-5, f"{name} encoder output")


@pytest.mark.parametrize("name", ALL)
def test_decode_step_vs_golden(name):
    d, cfg, _ = load(name)
    m, _ = model_for(name)
    enc = torch.from_numpy(d["g1_enc"]).to(DEV)
    tok = torch.full((4, 1), START, dtype=torch.long, device=DEV)
    hidden = None
    with torch.no_grad():
        for s in range(3):
            prev = None if hidden is None else (hidden[0].clone(), hidden[1].clone())
            logits, new_hidden = m.decoder.decode_step(enc, tok, hidden)
            if prev is not None:                     # inputs are never mutated (seq2seq.py:272 relies on it)
                assert torch.equal(prev[0], hidden[0]) and torch.equal(prev[1], hidden[1])
            hidden = new_hidden
            assert logits.shape == (4, 1, cfg["vocab_size"])
            close(logits.cpu().numpy(), d[f"g2_logits{s}"], 1e-4, f"{name} decode_step logits", absolute=True)
            close(hidden[0].cpu().numpy(), d[f"g2_h{s}"], 1e-5, f"{name} decode_step h")
            close(hidden[1].cpu().numpy(), d[f"g2_c{s}"], 1e-5, f"{name} decode_step c")
            tok = logits.squeeze(1).argmax(-1, keepdim=True)
            assert np.array_equal(tok.cpu().numpy(), d[f"g2_tok{s}"])


@pytest.mark.parametrize("name", ALL)
def test_greedy_ids_vs_golden(name):
    d, cfg, _ = load(name)
    m, _ = model_for(name)
    x = images(cfg, device=DEV)
    with torch.no_grad():
        ids = m.inference(x, START, END, max_length=32)
        one = m.inference(x[1:2], START, END, max_length=32)
        idt = m.inference(x, START, END, max_length=12, temperature=0.7)
        fb = m.inference(x, START, END, max_length=10, beam_size=3)          # batch > 1: greedy fallback
    assert np.array_equal(np.array(ids), d["g3_b4_ids"])
    assert list(one) == list(d["g3_b1_ids"])
    assert np.array_equal(np.array(idt), d["g3_b4_temp_ids"])
    assert np.array_equal(np.array(fb), d["g4_fallback_ids"])


@pytest.mark.parametrize("name", ALL)
def test_teacher_forced_logits_vs_golden(name):
    d, cfg, _ = load(name)
    m, _ = model_for(name)
    big = name in BIG
    T = 24 if big else 12
    forms = torch.from_numpy(synth.make_formulas(4, T, cfg["vocab_size"], seed=777, min_len=5)).to(DEV)
    with torch.no_grad():
        logits = m(images(cfg, device=DEV), forms)
    assert logits.shape == (4, T - 1, cfg["vocab_size"])
    if "g6_logits" in d:
        close(logits.cpu().numpy(), d["g6_logits"], 1e-4, f"{name} teacher-forced logits", absolute=True)
    else:
        close(sample(logits), d["g6_logits_sample"], 1e-4, f"{name} teacher-forced logits", absolute=True)


@pytest.mark.parametrize("name", ALL)
def test_sticky_stop_loop_vs_golden(name):
    """Predictor.predict_batch semantics (predictor.py:283-358)."""
    d, cfg, _ = load(name)
    m, _ = model_for(name)
    with torch.no_grad():
        enc = m.encoder(images(cfg, device=DEV))
        ids, _ = m.greedy_ids(enc, START, END, 32, stop=_lib.STOP_STICKY, select=_lib.SELECT_SOFTMAX)
    rows = []
    for r in ids.cpu().tolist():
        r = [t for t in r if t >= 0]
        rows.append(r[: r.index(END)] if END in r else r)
    assert rows == padded_to_lists(d["g5_ids"], d["g5_len"])


def _margin_guard(got, ref_ids, margins, tol):
    """ids must agree with the reference up to the first step whose reference top1-top2
    margin is below tol (a fp32 near-tie, after which sequences legitimately diverge)."""
    diverged = 0
    for b in range(ref_ids.shape[0]):
        ne = np.nonzero(got[b] != ref_ids[b, 1:1 + got.shape[1]])[0]
        if ne.size:
            t = int(ne[0])
            assert margins[b, t] < tol, f"row {b} step {t}: ids differ at margin {margins[b, t]}"
            diverged += 1
    return diverged


@pytest.mark.parametrize("fname", ["primary_cfg2_clock", "primary_cfg2"])
def test_cfg2_batch256_ids(fname):
    """BASELINE config 2 at full size: B=256, up to 150 steps."""
    d, cfg, sd_kw = load(fname)
    m, _ = model_for(fname, sd_kw, cfg)
    ref_ids = d["ids"].astype(np.int64)
    steps = ref_ids.shape[1] - 1
    x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(DEV)
    with torch.no_grad():
        enc = m.encoder(x)
        ids, _ = m.greedy_ids(enc, START, END, 150)
        seqs = m._greedy_search(enc, START, END, 150, 1.0, 0, 0.0)
        # teacher-forced replay along the REFERENCE ids: every step is checked independently
        forced = torch.from_numpy(ref_ids[:, :steps].astype(np.int32)).to(DEV)
        _, logits, _ = m.decoder.run_steps(enc, steps, forced[:, 0].contiguous(), forced=forced,
                                           want_ids=False, want_logits=True)
    close(enc.reshape(-1)[:: 256 * cfg["embedding_dim"] // 1024][:1024].cpu().numpy(), d["enc_sample"], 1e-5,
          f"{fname} B=256 encoder output")
    assert len(seqs[0]) == steps + 1, "global stop step (all rows END in one step) differs"
    got = ids.cpu().numpy()[:, :steps]
    diverged = _margin_guard(got, ref_ids, d["margins"], tol=2e-4)
    assert diverged <= 0.05 * 256
    top = logits.argmax(-1).cpu().numpy()
    wide = d["margins"] > 2e-4
    assert np.array_equal(top[wide], ref_ids[:, 1:][wide])
    top2 = torch.topk(logits, 2, dim=-1).values
    close((top2[..., 0] - top2[..., 1]).cpu().numpy(), d["margins"], 1e-4, f"{fname} B=256 top1-top2 margins", absolute=True)


def test_linear_kernel_odd_shapes():
    torch.manual_seed(0)
    L = _lib.lib()
    for (M, K, N, relu) in [(1, 7, 3, 0), (5, 33, 70, 1), (67, 130, 65, 1), (4, 4096, 33, 0), (256, 1024, 96, 1),
                            # long reductions: the split-bf16 matrix-core kernel (ragged tiles, K tail, full FC size)
                            (130, 2056, 70, 1), (256, 40960, 256, 1), (3, 8192, 200, 0)]:
        x, w, b = torch.randn(M, K), torch.randn(N, K) / K ** 0.5, torch.randn(N)
        want = torch.nn.functional.linear(x, w, b)
        want = want.relu() if relu else want
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
        y = torch.empty(M, N, device=DEV)
        nbytes = L.i2l_linear_workspace_bytes(M, K, N)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=DEV)
        rc = L.i2l_linear_bias_act_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), M, K, N, relu,
                                       ws.data_ptr(), nbytes, 0, _lib.stream_ptr())
        assert rc == 0
        close(y.cpu().numpy(), want.numpy(), 1e-5)


def test_conv_kernel_odd_shapes():
    torch.manual_seed(1)
    L = _lib.lib()
    for (B, Cin, H, W, Cout) in [(1, 1, 2, 2, 1), (2, 3, 7, 9, 5), (3, 5, 16, 70, 9), (2, 16, 33, 131, 24),
                                 (1, 64, 16, 80, 128), (2, 3, 64, 320, 32), (2, 5, 23, 71, 96), (3, 9, 7, 9, 32),
                                 (1, 32, 32, 160, 64), (2, 1, 2, 2, 64),
                                 # split-bf16 matrix-core kernel: every tile shape, ragged edges, 3 chunks
                                 (2, 16, 9, 13, 64), (2, 32, 25, 100, 64), (3, 64, 12, 50, 128), (1, 48, 37, 19, 192),
                                 (2, 16, 2, 2, 64),
                                 # first-block kernel (K = 9 Cin <= 27 in one or two bf16 k-steps)
                                 (2, 1, 50, 200, 32), (1, 2, 9, 13, 64), (3, 3, 17, 33, 32), (1, 3, 8, 64, 96)]:
        x, w, b = torch.randn(B, Cin, H, W), torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5), torch.randn(Cout)
        want = O.conv_block(x, w, b)
        y = torch.empty(B, Cout, H // 2, W // 2, device=DEV)
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)             # keep the device copies alive
        nbytes = L.i2l_conv_workspace_bytes(Cin, Cout)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=DEV)
        rc = L.i2l_conv3x3_relu_pool2_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(),
                                          y.data_ptr(), None, B, Cin, H, W, Cout, ws.data_ptr(), nbytes, 0,
                                          _lib.stream_ptr())
        assert rc == 0
        close(y.cpu().numpy(), want.numpy(), 1e-5)


def _conv_abi(x, w, b, flags=0):
    L = _lib.lib()
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y = torch.empty(B, Cout, H // 2, W // 2, device=DEV)
    nbytes = L.i2l_conv_workspace_bytes(Cin, Cout)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=DEV)
    assert L.i2l_conv3x3_relu_pool2_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), None, B, Cin, H, W, Cout,
                                        ws.data_ptr(), nbytes, flags, _lib.stream_ptr()) == 0
    return y.cpu()


def _linear_abi(x, w, b, flags=0):
    L = _lib.lib()
    (M, K), N = x.shape, w.shape[0]
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y = torch.empty(M, N, device=DEV)
    nbytes = L.i2l_linear_workspace_bytes(M, K, N)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=DEV)
    assert L.i2l_linear_bias_act_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), M, K, N, 0, ws.data_ptr(),
                                     nbytes, flags, _lib.stream_ptr()) == 0
    return y.cpu()


def _conv_truth(x, w, b):
    """float64 value of the block and the condition-aware scale sum |x||w| + |b| of every output (pooled: max of the 4)."""
    xd, wd, bd = x.double(), w.double(), b.double()
    pre = torch.nn.functional.conv2d(xd, wd, bd, padding=1)
    mag = torch.nn.functional.conv2d(xd.abs(), wd.abs(), bd.abs(), padding=1)
    return torch.nn.functional.max_pool2d(torch.relu(pre), 2), torch.nn.functional.max_pool2d(mag, 2)


ADVERSARIAL = ["cancellation", "range_2^+-60", "tiny_2^-100", "huge_2^+100"]


def _adversarial(kind, x, w, chan_dim_x, chan_dim_w, g):
    """Rewrites (x, w) along their reduction (input-channel / K) axis; returns the operands."""
    n = x.shape[chan_dim_x]
    if kind == "cancellation":          # consecutive reduction slots cancel to ~2^-12 of their size
        xe, xo = x.narrow(chan_dim_x, 0, n // 2 * 2).unfold(chan_dim_x, 2, 2).unbind(-1)
        we, wo = w.narrow(chan_dim_w, 0, n // 2 * 2).unfold(chan_dim_w, 2, 2).unbind(-1)
        xo.copy_(-xe * (1.0 + 2.0 ** -12))
        wo.copy_(we)
        x, w = x * 64.0, w * 64.0
    elif kind == "range_2^+-60":        # slot c scaled by 2^e_c in x and 2^-e_c in w: products stay O(1)
        e = torch.randint(-60, 61, (n,), generator=g).double()
        shape_x = [1] * x.dim(); shape_x[chan_dim_x] = n
        shape_w = [1] * w.dim(); shape_w[chan_dim_w] = n
        x = (x.double() * (2.0 ** e).reshape(shape_x)).float()
        w = (w.double() * (2.0 ** -e).reshape(shape_w)).float()
    elif kind == "tiny_2^-100":         # low split pieces at 2^-116: still normal bf16 numbers
        x = x * 2.0 ** -100
    elif kind == "huge_2^+100":
        x, w = x * 2.0 ** 100, w * 2.0 ** -20
    return x.contiguous(), w.contiguous()


@pytest.mark.parametrize("kind", ADVERSARIAL)
def test_bf16x3_adversarial_operands(kind):
    """The 3 x bf16 split products (conv blocks 2/3, the first-block kernel, the long-K linear) on operands chosen to
    break a scheme that only LOOKS fp32-grade on randn data: pairwise cancellation, 2^+-60 dynamic range inside one
    reduction, magnitudes near the ends of the exponent range.  Judged against float64 with the condition-aware
    bound  |err| <= 2^-21 * (sum |x||w| + |b|)  -- the class of an fp32 fmaf chain of this length -- and never worse
    than 4x the error of the fp32 CPU oracle on the same operands; the exact-fp32 kernels (I2L_FLAG_EXACT_FP32) must
    meet the same bound."""
    g = torch.Generator().manual_seed(ADVERSARIAL.index(kind) + 17)
    worst = {}
    for (B, Cin, H, W, Cout) in [(2, 32, 12, 20, 64), (1, 64, 8, 16, 128), (2, 3, 12, 40, 32)]:
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
        b = torch.randn(Cout, generator=g) * (0.0 if kind != "cancellation" else 1.0)
        if kind == "tiny_2^-100":
            b = b * 0.0
        x, w = _adversarial(kind, x, w, 1, 1, g)
        truth, mag = _conv_truth(x, w, b)
        oracle_err = (O.conv_block(x, w, b).double() - truth).abs()
        for flags in (0, _lib.FLAG_EXACT_FP32):
            got = _conv_abi(x, w, b, flags).double()
            assert torch.isfinite(got).all()
            ratio = float(((got - truth).abs() / (mag + 1e-300)).max())
            worst[(Cin, flags)] = ratio
            assert ratio <= 2.0 ** -21, (kind, Cin, flags, ratio)
            assert float(((got - truth).abs() - 4 * oracle_err - 2.0 ** -22 * mag).max()) <= 0, (kind, Cin, flags)
    for (M, K, N) in [(5, 4096, 70), (64, 2048, 33)]:
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.zeros(N)
        x, w = _adversarial(kind, x, w, 1, 1, g)
        truth = x.double() @ w.double().t()
        mag = x.double().abs() @ w.double().abs().t()
        for flags in (0, _lib.FLAG_EXACT_FP32):
            got = _linear_abi(x, w, b, flags).double()
            ratio = float(((got - truth).abs() / (mag + 1e-300)).max())
            worst[("linear", K, flags)] = ratio
            assert ratio <= 2.0 ** -20, (kind, K, flags, ratio)
    for k, v in worst.items():
        record(f"bf16x3 adversarial {kind} {k} [err / sum|x||w|]", v)


def test_bf16x3_documented_limits():
    """What the split scheme does NOT promise (include/img2latex_hip.h): (a) operands below ~2^-110 lose their low
    split pieces to bf16's subnormal range -- the error may grow to bf16 class (bounded here by 2^-7 of sum |x||w|)
    while the exact-fp32 kernels stay near the fp32 class; (b) non-finite inputs poison only the outputs whose receptive field holds them (every
    other output stays exactly what it was), but the poisoned value is unspecified -- NaN where the reference gives
    +-Inf, and ReLU's max may drop a NaN -- so callers must not rely on non-finite propagation."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 32, 8, 16, generator=g) * 2.0 ** -116
    w = torch.randn(64, 32, 3, 3, generator=g) * 2.0 ** 60
    b = torch.zeros(64)
    truth, mag = _conv_truth(x, w, b)
    split = float(((_conv_abi(x, w, b, 0).double() - truth).abs() / (mag + 1e-300)).max())
    exact = float(((_conv_abi(x, w, b, _lib.FLAG_EXACT_FP32).double() - truth).abs() / (mag + 1e-300)).max())
    assert split <= 2.0 ** -7 and exact <= 2.0 ** -18, (split, exact)
    record("bf16x3 subnormal-range operands, split kernel [err / sum|x||w|]", split)
    record("bf16x3 subnormal-range operands, exact kernel [err / sum|x||w|]", exact)
    x = torch.randn(2, 32, 12, 20, generator=g)
    w = torch.randn(64, 32, 3, 3, generator=g) / 17.0
    clean = _conv_abi(x, w, b, 0)
    bad = x.clone()
    bad[1, 5, 6, 9] = float("inf")
    bad[0, 0, 0, 0] = float("nan")
    got = _conv_abi(bad, w, b, 0)
    touched = torch.zeros(2, 1, 12, 20)
    touched[1, 0, 5:8, 8:11] = 1.0
    touched[0, 0, 0:2, 0:2] = 1.0
    touched = torch.nn.functional.max_pool2d(touched, 2).bool().expand_as(got)
    assert torch.equal(got[~touched], clean[~touched])


def test_error_codes_not_exceptions():
    L = _lib.lib()
    assert L.i2l_conv3x3_relu_pool2_fwd(None, None, None, None, None, 1, 1, 4, 4, 1, None, 0, 0, None) == -1
    assert L.i2l_linear_bias_act_fwd(None, None, None, None, 1, 1, 1, 0, None, 0, 0, None) == -1


def test_rows_are_independent_and_batch_tail():
    """Ragged batch sizes: B not a multiple of the rows-per-workgroup tiling; row b of a
    batch must equal the same image decoded alone."""
    d, cfg, _ = load("tiny_l2_attn")
    m, _ = model_for("tiny_l2_attn")
    x = images(cfg, batch=7, seed=99, device=DEV)
    with torch.no_grad():
        enc = m.encoder(x)
        ids7, _ = m.greedy_ids(enc, START, END, 20)
        ids1, _ = m.greedy_ids(enc[3:4].contiguous(), START, END, 20)
        big = enc.repeat(90, 1).contiguous()                     # 630 rows -> 4 rows per workgroup, ragged tail
        idsb, _ = m.greedy_ids(big, START, END, 20)
    assert torch.equal(ids7[3], ids1[0])
    assert torch.equal(idsb[: 7], ids7) and torch.equal(idsb[623:630], ids7)


@pytest.mark.parametrize("name", ALL)
def test_beam_search_vs_golden(name):
    """seq2seq.py:234-298: N independent batch-1 beam searches in one launch."""
    d, cfg, _ = load(name)
    m, _ = model_for(name)
    bimgs = images(cfg, 8, seed=4321, device=DEV)
    with torch.no_grad():
        enc = m.encoder(bimgs)
        for k in (5, 3):
            want = padded_to_lists(d[f"g4_k{k}_ids"], d[f"g4_k{k}_len"])
            got, scores = m.beam_search_batch(enc, START, END, 40, k, return_scores=True)
            assert got == want, (name, k)
            assert all(np.isfinite(s) and s <= 0.0 for s in scores)
        one = m.inference(bimgs[2:3], START, END, max_length=40, beam_size=5)   # the reference's entry point
    assert one == padded_to_lists(d["g4_k5_ids"], d["g4_k5_len"])[2]


@pytest.mark.parametrize("fname", ["primary_cfg3_beam", "primary_cfg3_beam_noend"])
@pytest.mark.parametrize("flags", [0, _lib.FLAG_NO_GROUP], ids=["grouped", "workgroup_per_image"])
def test_cfg3_beam_full_size_vs_reference(fname, flags):
    """BASELINE configs[2] at ITS size against the reference itself (seq2seq.py:234-298 run per image by
    make_golden.run_cfg3_beam): 128 images, k = 5, max_length 150, attention on, E = H = 256, V = 512 -- tokens and
    the winning beam's score, for beam_group_kernel and beam_kernel.  All but one of the 128 searches run the full 150
    steps (the best beam completes after 14-26 tokens, the others keep four live beams).  A search ranks 25 fp64
    sums of fp32 log-probabilities per step; the fixture records each image's smallest gap between neighbours among
    the 6 best candidates of any step, and an fp32 evaluation in another summation order may rank differently below
    ~1e-4 (150 steps x the ~5e-6 logit error).  So: tokens exact for every image whose smallest gap is >= 1e-4;
    images below it may differ and are counted (measured: none do); scores <= 1e-4 relative for every equal sequence."""
    d, cfg, sd_kw = load(fname)
    m, _ = model_for(fname, sd_kw, cfg)
    n, k, T = len(d["lens"]), int(d["k"]), int(d["max_length"])
    x = torch.from_numpy(synth.make_images(n, cfg, seed=int(d["image_seed"]))).to(DEV)
    with torch.no_grad():
        enc = m.encoder(x)
        got, scores = m.beam_search_batch(enc, START, END, T, k, return_scores=True, flags=flags)
        one = m.inference(x[3:4], START, END, max_length=T, beam_size=k)          # the reference's entry point
    want = padded_to_lists(d["ids"].astype(np.int64), d["lens"])
    assert one == want[3]
    off, worst = [], 0.0
    for j in range(n):
        if got[j] != want[j]:
            assert d["min_gap"][j] < 1e-4, (fname, j, float(d["min_gap"][j]))
            off.append(j)
            continue
        err = abs(scores[j] - float(d["scores"][j])) / max(1.0, abs(float(d["scores"][j])))
        worst = max(worst, err)
        assert err <= 1e-4, (fname, j, scores[j], float(d["scores"][j]))
    record(f"{fname} {n} images x k={k} x {T} steps [{'grouped' if flags == 0 else 'workgroup per image'}]: images whose "
           f"tokens leave the reference's at a near-tie (of {int((d['min_gap'] < 1e-4).sum())} below the guard)", len(off))
    record(f"{fname} winning scores vs the reference [rel to max(1,|score|)]", worst)
    assert len(off) <= 0.05 * n, off


@pytest.mark.parametrize("k", [2, 3, 4, 5, 6])
def test_grouped_beam_matches_workgroup_per_image(k):
    """beam_group_kernel (4 workgroups share 12 beam slots, weights on chip, per-step exchanges) against beam_kernel
    (one workgroup per image) at the primary dimensions: same sequences, scores within fp32 log-softmax rounding.
    37 images: several groups per launch and a last group with unused image slots."""
    d, cfg, _ = load("primary")
    m, _ = model_for("primary")
    bimgs = images(cfg, 37, seed=99 + k, device=DEV)
    with torch.no_grad():
        enc = m.encoder(bimgs)
        got, gs = m.beam_search_batch(enc, START, END, 48, k, return_scores=True)
        want, ws = m.beam_search_batch(enc, START, END, 48, k, return_scores=True, flags=_lib.FLAG_NO_GROUP)
    assert got == want
    assert np.allclose(gs, ws, rtol=1e-5, atol=1e-4)


def test_grouped_beam_small_vocabulary_and_many_groups():
    """Vocabulary 300 (< 512: member 2 of a group owns 44 valid columns, member 3 none, so its candidates are all
    invalid and its (max, sum exp) is (-inf, 0)); 140 images at k = 5 = 70 groups = 280 workgroups, more than the
    chip holds at once, so later groups start as earlier ones finish."""
    cfg = synth.model_config(vocab_size=300, attention=True)
    m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
    sd = synth.make_state_dict(cfg, seed=11, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(DEV).eval()
    imgs = torch.from_numpy(synth.make_images(140, cfg, seed=5)).to(DEV)
    with torch.no_grad():
        enc = m.encoder(imgs)
        got, gs = m.beam_search_batch(enc, START, END, 40, 5, return_scores=True)
        want, ws = m.beam_search_batch(enc, START, END, 40, 5, return_scores=True, flags=_lib.FLAG_NO_GROUP)
    assert got == want
    assert np.allclose(gs, ws, rtol=1e-5, atol=1e-4)
    assert max(max(s) for s in got if s) < 300


def test_beam_scores_vs_oracle():
    d, cfg, _ = load("tiny_l2_attn")
    m, _ = model_for("tiny_l2_attn")
    sd = torch_state_dict("tiny_l2_attn")
    bimgs = images(cfg, 6, seed=77)
    with torch.no_grad():
        enc_cpu = O.cnn_encoder(sd, cfg, bimgs)
        got, scores = m.beam_search_batch(m.encoder(bimgs.to(DEV)), START, END, 30, 4, return_scores=True)
        for j in range(6):
            seq, sc = O.beam_search(sd, cfg, enc_cpu[j:j + 1], START, END, 30, 4, return_score=True)
            assert got[j] == seq
            assert abs(scores[j] - sc) <= 1e-4 * max(1.0, abs(sc))
        # k = 1 beam == greedy until END; max beam width; START == END degenerate case
        g1, _ = m.beam_search_batch(m.encoder(bimgs.to(DEV)), START, END, 30, 1, return_scores=True)
        for j in range(6):
            assert g1[j] == O.beam_search(sd, cfg, enc_cpu[j:j + 1], START, END, 30, 1)
        g8 = m.beam_search_batch(m.encoder(bimgs.to(DEV)), START, END, 12, 8)
        for j in range(2):
            assert g8[j] == O.beam_search(sd, cfg, enc_cpu[j:j + 1], START, END, 12, 8)
        assert m.beam_search_batch(m.encoder(bimgs.to(DEV)), START, START, 5, 3)[0] == []


def test_attention_general_length():
    from img2latex_amd.model import Attention
    torch.manual_seed(3)
    for (B, S, H, E) in [(3, 1, 64, 32), (4, 7, 64, 32), (2, 33, 128, 96)]:
        att = Attention(H, E)
        sd = {"decoder.attention.attn.weight": att.attn.weight.detach().clone(),
              "decoder.attention.attn.bias": att.attn.bias.detach().clone(),
              "decoder.attention.v.weight": att.v.weight.detach().clone()}
        hid, enc = torch.randn(B, 1, H), torch.randn(B, S, E)
        with torch.no_grad():
            want = O.attention_context(sd, hid, enc)
            got = att.to(DEV)(hid.to(DEV), enc.to(DEV)).cpu()
        close(got.numpy(), want.numpy(), 1e-5)
        if S == 1:                                   # identity, bit for bit (fixture G7)
            assert torch.equal(got, enc)


def test_greedy_pipeline_matches_serial():
    """Two-stream batch pipeline returns exactly what the serial search returns, batch by batch."""
    from img2latex_amd.pipeline import GreedyPipeline
    d, cfg, _ = load("tiny_l1")
    m, _ = model_for("tiny_l1")
    batches = [images(cfg, batch=5, seed=100 + i, device=DEV) for i in range(6)]
    with torch.no_grad():
        want = [m.inference(b, START, END, max_length=32) for b in batches]
    pipe = GreedyPipeline(m, START, END, 32, depth=2)
    got = []
    for b in batches:
        if pipe.pending() >= 2:
            got.append(pipe.to_sequences(pipe.collect()))
        pipe.submit(b)
    while pipe.pending():
        got.append(pipe.to_sequences(pipe.collect()))
    assert got == want
    with pytest.raises(RuntimeError):
        for b in batches[:3]:
            pipe.submit(b)
    pipe.drain()


# (B, H, W, Cin, Cout, k, stride, pad, residual, nchw fp32 images): every bf16 conv path of the ResNet trunk --
# direct 1x1 and implicit-GEMM 3x3 / strided 1x1 through the ring-buffered kernel (2 stages below 4 K tiles,
# 4 stages from there; 64- and 128-column tiles; ragged M and N), the fused 7x7 stem with ragged tiles, and the
# im2col fallbacks (Cin % 64 != 0, unusual stem)
BF16_CONV_CASES = [
    (2, 9, 13, 64, 64, 3, 1, 1, 0, 0), (3, 8, 10, 64, 256, 1, 1, 0, 1, 0), (2, 7, 9, 256, 64, 1, 1, 0, 0, 0),
    (2, 10, 14, 128, 128, 3, 2, 1, 0, 0), (2, 10, 14, 256, 512, 1, 2, 0, 0, 0), (1, 5, 7, 512, 512, 3, 1, 1, 1, 0),
    (5, 6, 6, 128, 72, 1, 1, 0, 1, 0), (2, 6, 6, 64, 200, 3, 1, 1, 1, 0), (1, 4, 5, 2048, 512, 1, 1, 0, 0, 0),
    (3, 18, 75, 3, 64, 7, 2, 3, 0, 1), (2, 64, 320, 3, 64, 7, 2, 3, 0, 1), (2, 12, 12, 3, 32, 7, 2, 3, 0, 1),
    (2, 9, 9, 32, 64, 3, 1, 1, 1, 0),
    # 3x3 / stride 1 on the LDS-patch kernel (csrc/resnet_patch.inc.h): tile rows that straddle images (H < rows per tile),
    # a grid that is not a multiple of the tile, wide rows (one row per tile), residual, 64- and 128-channel tiles
    (12, 2, 10, 128, 128, 3, 1, 1, 0, 0), (3, 4, 20, 64, 256, 3, 1, 1, 1, 0), (2, 16, 80, 64, 64, 3, 1, 1, 0, 0),
    (1, 8, 100, 64, 128, 3, 1, 1, 0, 0), (5, 3, 7, 64, 128, 3, 1, 1, 1, 0), (1, 3, 200, 64, 64, 3, 1, 1, 0, 0),
    (7, 1, 13, 128, 64, 3, 1, 1, 0, 0),
    # 320 tiles of 128 x 128: the balance rule takes 64-column tiles (I2L_FLAG_RESNET_WIDE_TILES keeps the wide ones)
    (16, 16, 80, 64, 256, 1, 1, 0, 0, 0),
]


@pytest.mark.parametrize("case", BF16_CONV_CASES)
def test_bf16_conv_bn_act_vs_torch(case):
    """i2l_conv_bn_act_bf16_fwd against fp32 torch on the same bf16-rounded operands (encoder.py:132-249 trunk ops):
    the only differences are fp32 summation order and the final rounding to bf16 (2^-8 relative)."""
    B, H, W, Cin, Cout, k, s, pd, res, nchw = case
    L = _lib.lib()
    g = torch.Generator().manual_seed(sum(case))
    w = torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    mean, var = torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5
    x = torch.randn(B, Cin, H, W, generator=g)
    Ho, Wo = (H + 2 * pd - k) // s + 1, (W + 2 * pd - k) // s + 1
    r = torch.randn(B, Cout, Ho, Wo, generator=g) if res else None
    rnd = lambda t: t.to(torch.bfloat16).to(torch.float32)
    scale = gamma / torch.sqrt(var + 1e-5)
    want = torch.nn.functional.conv2d(rnd(x), rnd(w), stride=s, padding=pd) * scale[None, :, None, None] \
        + (beta - mean * scale)[None, :, None, None]
    if res:
        want = rnd(want) + rnd(r)           # the kernel rounds the scaled tile once before adding the residual
    want = torch.relu(want)
    nb = L.i2l_conv_bf16_packed_bytes(Cout, Cin, k, k)
    packed = torch.empty(nb, dtype=torch.uint8, device=DEV)
    dv = lambda t: t.contiguous().to(DEV)
    wd, gd, bd, md, vd = dv(w), dv(gamma), dv(beta), dv(mean), dv(var)
    _lib.check(L.i2l_conv_bn_bf16_pack(wd.data_ptr(), gd.data_ptr(), bd.data_ptr(), md.data_ptr(), vd.data_ptr(), 1e-5,
                                       packed.data_ptr(), nb, Cout, Cin, k, k, _lib.stream_ptr()), "pack")
    xd = dv(x) if nchw else dv(x.permute(0, 2, 3, 1)).to(torch.bfloat16)
    rd = dv(r.permute(0, 2, 3, 1)).to(torch.bfloat16) if res else None
    y = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    outs = []
    # every kernel behind the entry point: automatic choice, single-buffered GEMM, forced ring depths, im2col stem
    variants = [0, _lib.FLAG_RESNET_NO_RING, _lib.flag_resnet_ring_depth(2), _lib.flag_resnet_ring_depth(3),
                _lib.flag_resnet_ring_depth(4), _lib.flag_resnet_ring_depth(5), _lib.FLAG_RESNET_IM2COL_STEM,
                _lib.FLAG_RESNET_WIDE_TILES]
    if k == 3 and s == 1 and pd == 1:     # the patch kernel's five tile shapes (a shape that does not fit falls back) and its switch
        variants += [_lib.flag_resnet_patch_shape(n) for n in range(1, 6)] + [_lib.FLAG_RESNET_NO_PATCH]
    for flags in variants:
        y.fill_(float("nan"))
        wsb = L.i2l_conv_bf16_workspace_bytes(B, H, W, Cin, Cout, k, k, s, pd, flags)
        ws = torch.empty(max(wsb, 256), dtype=torch.uint8, device=DEV)
        _lib.check(L.i2l_conv_bn_act_bf16_fwd(xd.data_ptr(), nchw, packed.data_ptr(), _lib.ptr(rd), y.data_ptr(), B, H, W, Cin,
                                              Cout, k, k, s, pd, 1, ws.data_ptr(), wsb, flags, _lib.stream_ptr()),
                   "conv_bn_act_bf16_fwd")
        got = y.float().cpu().permute(0, 3, 1, 2)
        assert torch.isfinite(got).all()
        err = (got - want).abs()
        assert float((err - 2.0 ** -7 * want.abs()).max()) <= 2e-3, (flags, float(err.max()))
        outs.append(got)
    for o in outs[1:]:       # the variants differ by fp32 summation order only: at most one bf16 ulp apart
        assert float(((o - outs[0]).abs() - 2.0 ** -7 * outs[0].abs()).max()) <= 1e-3


@pytest.mark.parametrize("n2,shape", [(64, (3, 5, 20)), (128, (2, 16, 80)), (64, (1, 1, 1)), (128, (5, 7, 37))])
def test_bottleneck_join_equals_two_launches(n2, shape):
    """i2l_bottleneck_join_bf16_fwd (conv3 + bn3 + identity + ReLU, then the next block's conv1 + bn1 + ReLU, the block output's
    tile never read back from HBM; encoder.py:185-249) against the two i2l_conv_bn_act_bf16_fwd launches it replaces: the same
    roundings in the same order, so y and z are bit-identical -- ragged position counts, both head widths; other shapes are
    refused with I2L_ERR_UNSUPPORTED so that the caller falls back to two launches."""
    B, H, W = shape
    L = _lib.lib()
    g = torch.Generator().manual_seed(100 + n2 + B * H * W)
    dv = lambda t: t.contiguous().to(DEV)
    o2 = dv((torch.randn(B, H, W, 64, generator=g) * 0.7).to(torch.bfloat16))
    ident = dv(torch.randn(B, H, W, 256, generator=g).to(torch.bfloat16))

    def packed(cout, cin):
        w = dv(torch.randn(cout, cin, 1, 1, generator=g) * cin ** -0.5)
        gam, bet = dv(torch.rand(cout, generator=g) + 0.5), dv(torch.randn(cout, generator=g) * 0.1)
        mean, var = dv(torch.randn(cout, generator=g) * 0.1), dv(torch.rand(cout, generator=g) + 0.5)
        nb = L.i2l_conv_bf16_packed_bytes(cout, cin, 1, 1)
        buf = torch.empty(nb, dtype=torch.uint8, device=DEV)
        _lib.check(L.i2l_conv_bn_bf16_pack(w.data_ptr(), gam.data_ptr(), bet.data_ptr(), mean.data_ptr(), var.data_ptr(), 1e-5,
                                           buf.data_ptr(), nb, cout, cin, 1, 1, _lib.stream_ptr()), "pack")
        return buf
    p3, p1 = packed(256, 64), packed(n2, 256)
    ws = torch.empty(4096, dtype=torch.uint8, device=DEV)
    y_ref = torch.empty(B, H, W, 256, dtype=torch.bfloat16, device=DEV)
    z_ref = torch.empty(B, H, W, n2, dtype=torch.bfloat16, device=DEV)
    for flags in (0, _lib.FLAG_RESNET_WIDE_TILES, _lib.FLAG_RESNET_NO_RING):     # whatever tile the two launches take
        _lib.check(L.i2l_conv_bn_act_bf16_fwd(o2.data_ptr(), 0, p3.data_ptr(), ident.data_ptr(), y_ref.data_ptr(), B, H, W, 64, 256,
                                              1, 1, 1, 0, 1, ws.data_ptr(), 4096, flags, _lib.stream_ptr()), "conv3")
        _lib.check(L.i2l_conv_bn_act_bf16_fwd(y_ref.data_ptr(), 0, p1.data_ptr(), None, z_ref.data_ptr(), B, H, W, 256, n2,
                                              1, 1, 1, 0, 1, ws.data_ptr(), 4096, flags, _lib.stream_ptr()), "conv1")
        y = torch.full_like(y_ref, float("nan"))
        z = torch.full_like(z_ref, float("nan"))
        _lib.check(L.i2l_bottleneck_join_bf16_fwd(o2.data_ptr(), p3.data_ptr(), ident.data_ptr(), y.data_ptr(), p1.data_ptr(),
                                                  z.data_ptr(), B * H * W, 64, 256, n2, _lib.stream_ptr()), "join")
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16)), flags
        assert torch.equal(z.view(torch.int16), z_ref.view(torch.int16)), flags
    for bad in ((128, 512, 128), (64, 256, 256), (64, 128, 64)):
        assert L.i2l_bottleneck_join_bf16_fwd(o2.data_ptr(), p3.data_ptr(), ident.data_ptr(), y.data_ptr(), p1.data_ptr(),
                                              z.data_ptr(), B * H * W, bad[0], bad[1], bad[2], _lib.stream_ptr()) == -2
    assert L.i2l_bottleneck_join_bf16_fwd(None, p3.data_ptr(), ident.data_ptr(), y.data_ptr(), p1.data_ptr(), z.data_ptr(),
                                          B * H * W, 64, 256, n2, _lib.stream_ptr()) == -1


@pytest.mark.parametrize("model_name,hw", [("resnet18", (32, 64)), ("resnet50", (64, 96)), ("resnet34", (32, 64)),
                                           ("resnet101", (32, 96)), ("resnet152", (32, 64))])
def test_resnet_encoder_vs_oracle(model_name, hw):
    """bf16 MFMA trunk against the fp32 restatement (parity unpinned: no reference output exists), every model name
    the reference accepts (encoder.py:185-196); measured 6e-3 .. 1.1e-2 of the feature maximum."""
    import resnet_oracle as RO
    from img2latex_amd.model import ResNetEncoder
    enc = ResNetEncoder(hw[0], hw[1], 3, model_name=model_name, embedding_dim=64)
    shapes = [(k, tuple(v.shape)) for k, v in enc.state_dict().items()]
    np_sd = synth.make_resnet_state_dict(shapes, seed=5)
    enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in np_sd.items()}, strict=True)
    enc = enc.to(DEV).eval()
    x = torch.from_numpy(synth.uniform(9, "rimg", (3, 3, hw[0], hw[1]), -1.0, 1.0))
    sd = {"encoder." + k: torch.from_numpy(v) for k, v in np_sd.items()}
    with torch.no_grad():
        want_feat = RO.resnet_trunk(sd, model_name, x)
        want = RO.resnet_encoder(sd, model_name, x)
        got_feat = enc.trunk(x.to(DEV)).cpu()
        got = enc(x.to(DEV)).cpu()
    assert got.shape == (3, 64)
    # bf16 activations through 18-50 layers: a few 1e-2 relative to the feature scale
    tol = 4e-2
    e_feat = float((got_feat - want_feat).abs().max()) / float(want_feat.abs().max())
    e_out = float((got - want).abs().max()) / max(1.0, float(want.abs().max()))
    record(f"{model_name} {hw[0]}x{hw[1]} trunk features vs fp32 oracle [rel to max]", e_feat)
    assert e_feat <= tol and e_out <= tol, (e_feat, e_out)
    enc.train()                                       # training mode: batch statistics (tests/test_resnet_training.py)
    with torch.no_grad():
        tr = enc(x.to(DEV))
    assert tr.shape == (3, 64) and torch.isfinite(tr).all() and not torch.equal(tr.cpu(), got)
    enc.eval()


def test_predict_ids_stream_equals_predict_batch_ids():
    """Predictor.predict_ids_stream (GreedyPipeline with the Predictor's loop: argmax of softmax, sticky stop): batch by batch
    the lists predict_batch_ids returns -- on a tiny attention model (no grouped kernels: the pipeline's C call picks the
    row-per-workgroup kernel) and at the primary dims (8-member co-resident decode; a row may leave the 4-member kernel's
    ids only at an fp32 near-tie, judged by the reference fixture's margins); ragged last batch, order preserved."""
    from img2latex_amd.training import Predictor, TokenTable
    for name, n_imgs, bs, T in (("tiny_l2_attn", 10, 4, 24), ("primary_cfg2_clock", 160, 64, 60)):
        d, cfg, sd_kw = load(name)
        m, _ = model_for(name, sd_kw, cfg) if name != "tiny_l2_attn" else model_for(name)
        vocab = {"<PAD>": 0, "<START>": 1, "<END>": 2, "<UNK>": 3}
        vocab.update({f"t{i}": i for i in range(4, cfg["vocab_size"])})
        pred = Predictor(m, TokenTable(vocab, max_sequence_length=150), device=torch.device(DEV))
        x = torch.from_numpy(synth.make_images(n_imgs, cfg, seed=1234)).to(DEV)
        batches = [x[i:i + bs] for i in range(0, n_imgs, bs)]
        want = [pred.predict_batch_ids(b, max_length=T) for b in batches]
        got = list(pred.predict_ids_stream(iter(batches), max_length=T))
        assert [len(g) for g in got] == [len(w) for w in want]
        differ = sum(1 for g, w in zip(got, want) for a, b in zip(g, w) if a != b)
        assert differ <= (0 if name == "tiny_l2_attn" else 2), (name, differ)


def test_predictor_and_checkpoint_roundtrip(tmp_path):
    """Checkpoint dict of trainer.py:209-224 -> Predictor.from_checkpoint (predictor.py:61-137) -> strings."""
    from img2latex_amd.training import Predictor, TokenTable, TrainStep, save_checkpoint
    d, cfg, sd_kw = load("tiny_l2_attn")
    m, _ = model_for("tiny_l2_attn")
    vocab = {"<PAD>": 0, "<START>": 1, "<END>": 2, "<UNK>": 3}
    vocab.update({f"t{i}": i for i in range(4, cfg["vocab_size"])})
    tok = TokenTable(vocab, max_sequence_length=150)
    config = {"model": {"name": "cnn_lstm", "embedding_dim": cfg["embedding_dim"],
                        "encoder": {"cnn": {k: v for k, v in synth.encoder_params(cfg).items() if k != "embedding_dim"}},
                        "decoder": synth.decoder_params(cfg)}}
    path = str(tmp_path / "best_checkpoint.pt")
    save_checkpoint(path, m, tok, config, epoch=3, step=77)
    ck = torch.load(path, weights_only=False)
    assert set(ck) >= {"epoch", "step", "model_state_dict", "optimizer_state_dict", "metrics", "config", "tokenizer_config"}
    pred = Predictor.from_checkpoint(path, device=torch.device(DEV))
    x = images(cfg, device=DEV)
    rows = pred.predict_batch_ids(x, max_length=32)
    assert [r[1:] for r in rows] == padded_to_lists(d["g5_ids"], d["g5_len"])          # fixture G5 (START stripped)
    # the fixture's tensors are sized for THIS model (16x32), not for the 64x800 `_prepare_image` hard-codes: bypass it
    # exactly as make_golden.py does to the reference's Predictor (tests/test_predict_chain.py covers the real thing)
    pred._prepare_image = lambda im: im.unsqueeze(0)
    texts = pred.predict_batch([x[i] for i in range(4)], max_length=32, batch_size=3)
    assert texts == [" ".join(f"t{t}" for t in r[1:] if t > 3) for r in rows]
    one = pred.predict(x[1], max_length=32)
    assert one == " ".join(f"t{t}" for t in d["g3_b1_ids"] if t > 3)
    del pred._prepare_image
    with pytest.raises(TypeError):
        pred.predict_batch([3.5])
    # optimizer state in torch.optim.Adam's layout, loadable by the real optimizer
    m2, _ = model_for("tiny_l1")
    ts = TrainStep(m2)
    forms = torch.from_numpy(synth.make_formulas(4, 10, 50, seed=1, min_len=4)).to(DEV)
    ts.step(images(load("tiny_l1")[1], device=DEV), forms)
    osd = ts.optimizer_state_dict()
    opt = torch.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=1e-4)
    opt.load_state_dict(osd)
    assert opt.state_dict()["state"][0]["exp_avg"].shape == m2.encoder.cnn_layers[0].weight.shape
    m2.eval()
    del _MODELS[("tiny_l1", repr(None))]            # its parameters were updated by the step above


def _uniform01(seed, row, step):
    """Python twin of decode.hip::uniform01 (splitmix64 finaliser of (seed, row, step), 24 bits)."""
    M = (1 << 64) - 1
    z = (seed + (((row << 32) | step) * 0x9E3779B97F4A7C15)) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z = z ^ (z >> 31)
    return np.float32(z >> 40) * np.float32(1.0 / 16777216.0)


@pytest.mark.parametrize("top_k,top_p,temp", [(5, 0.0, 1.0), (0, 0.8, 0.7), (7, 0.6, 1.3), (1, 0.0, 1.0)])
def test_sampling_distribution_and_draws(top_k, top_p, temp):
    """predictor.py:295-331: the masked / renormalised distribution equals the oracle's; every draw is the
    inverse-CDF of that distribution at the kernel's documented uniform; same seed -> same ids."""
    d, cfg, _ = load("tiny_l2_attn")
    m, _ = model_for("tiny_l2_attn")
    sd = torch_state_dict("tiny_l2_attn")
    x = images(cfg, batch=6, seed=41)
    seed, steps = 1234567, 6
    with torch.no_grad():
        enc = m.encoder(x.to(DEV))
        tok0 = torch.full((6,), START, dtype=torch.int32, device=DEV)
        ids, probs = m.decoder.sample_steps(enc, steps, tok0, temp, top_k, top_p, seed, stop=_lib.STOP_NONE, want_probs=True)
        ids2, _ = m.decoder.sample_steps(enc, steps, tok0, temp, top_k, top_p, seed, stop=_lib.STOP_NONE)
        ids3, _ = m.decoder.sample_steps(enc, steps, tok0, temp, top_k, top_p, seed + 1, stop=_lib.STOP_NONE)
    assert torch.equal(ids, ids2)
    ids_h, probs_h = ids.cpu().numpy(), probs.cpu().numpy()
    # oracle: replay the decoder along the HIP-sampled tokens and rebuild the masked distribution per step
    enc_cpu = O.cnn_encoder(sd, cfg, x)
    tok = torch.full((6, 1), START, dtype=torch.long)
    hidden = None
    for t in range(steps):
        with torch.no_grad():
            out, hidden = O.decode_step(sd, cfg, enc_cpu, tok, hidden)
        p = torch.softmax(out.squeeze(1) / temp, dim=-1)
        if top_k > 0:
            kth = torch.topk(p, min(top_k, p.size(-1)), dim=-1).values[:, -1, None]
            p = torch.where(p < kth, torch.zeros_like(p), p)
            p = p / p.sum(-1, keepdim=True)
        if top_p > 0:
            sp, si = torch.sort(p, descending=True, stable=True)
            cum = torch.cumsum(sp, -1)
            rm = cum > top_p
            rm[:, 1:] = rm[:, :-1].clone()
            rm[:, 0] = False
            p = torch.where(rm.scatter(-1, si, rm), torch.zeros_like(p), p)
            p = p / p.sum(-1, keepdim=True)
        want = p.numpy()
        close(probs_h[:, t, :], want, 1e-5)
        for b in range(6):
            pr = probs_h[b, t].astype(np.float32)
            assert pr[ids_h[b, t]] > 0
            u = _uniform01(seed, b, t)
            cdf = np.cumsum(pr.astype(np.float64))
            j = int(np.searchsorted(cdf, float(u) * cdf[-1], side="right"))
            lo = cdf[j - 1] if j > 0 else 0.0
            if min(abs(float(u) * cdf[-1] - lo), abs(cdf[min(j, len(cdf) - 1)] - float(u) * cdf[-1])) > 1e-5:
                assert j == ids_h[b, t], (b, t, j, ids_h[b, t])
        tok = torch.from_numpy(ids_h[:, t:t + 1].astype(np.int64))
    if top_k != 1:
        assert not torch.equal(ids, ids3)        # a different seed draws different tokens
    else:
        assert torch.equal(ids, ids3)            # top-1 sampling is greedy


def test_full_size_properties_cfg2():
    """Size-independent properties at BASELINE config 2's full size (B=256, 150 steps): idempotence
    (bit-identical reruns), batch-permutation equivariance (rows are independent), prefix property
    (the first k steps of a longer decode equal a k-step decode), sticky-stop consistency."""
    d, cfg, sd_kw = load("primary_cfg2_clock")
    m, _ = model_for("primary_cfg2_clock", sd_kw, cfg)
    x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(DEV)
    perm = torch.from_numpy(synth.randint(3, "perm", (256,), 0, 1 << 30)).argsort().to(DEV)
    with torch.no_grad():
        enc = m.encoder(x)
        enc2 = m.encoder(x)
        ids, _ = m.greedy_ids(enc, START, END, 150)
        ids_again, _ = m.greedy_ids(enc, START, END, 150)
        enc_p = m.encoder(x[perm].contiguous())
        ids_p, _ = m.greedy_ids(enc_p, START, END, 150)
        ids_40, _ = m.greedy_ids(enc, START, END, 40)
        ids_sticky, _ = m.greedy_ids(enc, START, END, 150, stop=_lib.STOP_STICKY)
    assert torch.equal(enc, enc2) and torch.equal(ids, ids_again)
    assert torch.equal(enc_p, enc[perm]) and torch.equal(ids_p, ids[perm])
    assert torch.equal(ids_40, ids[:, :40])
    a, s_ = ids.cpu().numpy(), ids_sticky.cpu().numpy()
    for b in range(256):
        ends = np.nonzero(a[b] == END)[0]
        n = int(ends[0]) + 1 if ends.size else 150
        assert np.array_equal(s_[b, :n], a[b, :n]) and (s_[b, n:] == -1).all()
    # ids are valid token ids and START never reappears as an argmax artefact of padding
    assert a.min() >= 0 and a.max() < cfg["vocab_size"]


def test_rows_per_workgroup_does_not_change_results():
    """The decode result must not depend on how rows are packed into workgroups (R = 1, 2, 4; resident-weight
    fast paths for R = 1 and R = 2)."""
    d, cfg, sd_kw = load("primary_cfg2")
    m, _ = model_for("primary_cfg2", sd_kw, cfg)
    x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)[:70]).to(DEV)
    with torch.no_grad():
        enc = m.encoder(x)
        base, _ = m.greedy_ids(enc, START, END, 60, rows_per_workgroup=1)
        for r in (2, 4):
            got, _ = m.greedy_ids(enc, START, END, 60, rows_per_workgroup=r)
            assert torch.equal(got, base), r
    assert np.array_equal(base.cpu().numpy(), d["ids"][:70, 1:61])


@pytest.mark.parametrize("rows", [1, 3, 4, 70, 256])
def test_grouped_decode_matches_row_per_workgroup_kernel(rows):
    """The grouped kernel (4 workgroups x 4 rows, weights on chip, in-launch exchanges; csrc/decode_group.inc.h) against
    the row-per-workgroup kernel on the same inputs: full and ragged groups, free-running ids (equal wherever the
    top-2 margin is not a rounding tie), teacher-forced logits, and the sticky stop rule."""
    d, cfg, sd_kw = load("primary_cfg2_clock")
    m, _ = model_for("primary_cfg2_clock", sd_kw, cfg)
    x = torch.from_numpy(synth.make_images(256, cfg, seed=77)[:rows]).to(DEV)
    T = 40
    with torch.no_grad():
        enc = m.encoder(x)
        tok0 = torch.full((rows,), START, dtype=torch.int32, device=DEV)
        # teacher forcing: identical inputs at every step -> logits comparable step by step
        forced = torch.from_numpy(synth.randint(5, "forced", (rows, T), 4, cfg["vocab_size"]).astype(np.int32)).to(DEV)
        _, lg_old, _ = m.decoder.run_steps(enc, T, tok0, forced=forced, want_logits=True, rows_per_workgroup=1)
        ids_grp, lg_grp, _ = m.decoder.run_steps(enc, T, tok0, forced=forced, want_logits=True)
        close(lg_grp.cpu().numpy(), lg_old.cpu().numpy(), 2e-5)
        assert np.array_equal(ids_grp.cpu().numpy(), lg_grp.argmax(-1).cpu().numpy())
        # free running, sticky stop: same ids except after a near-tie (then the sequences legitimately diverge)
        ids_old, lg_free, _ = m.decoder.run_steps(enc, T, tok0, stop=_lib.STOP_STICKY, end_id=END, want_logits=True,
                                                  rows_per_workgroup=1)
        ids_new, _, _ = m.decoder.run_steps(enc, T, tok0, stop=_lib.STOP_STICKY, end_id=END)
    a, b = ids_old.cpu().numpy(), ids_new.cpu().numpy()
    top2 = torch.topk(lg_free, 2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1]).cpu().numpy()
    for r in range(rows):
        diff = np.nonzero(a[r] != b[r])[0]
        if diff.size:
            t0 = int(diff[0])
            assert margin[r, t0] < 2e-4, (r, t0, margin[r, t0])
    assert (a == b).mean() > 0.97


def test_grouped_decode_under_concurrent_load():
    """The in-launch exchanges of the grouped kernel must not depend on timing or on having the GPU to itself:
    (a) decode while another stream keeps the CUs busy with conv launches (the group members start at different
    times and wait for each other), (b) two grouped decodes on two streams at once (each sees roughly half of the
    CUs; in-order dispatch keeps complete groups running).  Ids must equal the quiet run, no poll may time out."""
    d, cfg, sd_kw = load("primary_cfg2_clock")
    m, _ = model_for("primary_cfg2_clock", sd_kw, cfg)
    m2, _ = model_for("primary_cfg2_clock", sd_kw, cfg)
    x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(DEV)
    T = 150
    with torch.no_grad():
        enc = m.encoder(x)
        quiet, _ = m.greedy_ids(enc, START, END, T)
        quiet = _lib.check_ids(quiet.cpu())
        side, main = torch.cuda.Stream(), torch.cuda.current_stream()
        for rep in range(3):
            side.wait_stream(main)
            with torch.cuda.stream(side):                   # (a) encoder traffic next to the decode
                for _ in range(4):
                    m2.encoder(x)
            busy, _ = m.greedy_ids(enc, START, END, T)
            with torch.cuda.stream(side):                   # (b) a second grouped decode (own model = own workspace)
                other, _ = m2.greedy_ids(enc, START, END, T)
            again, _ = m.greedy_ids(enc, START, END, T)
            main.wait_stream(side)
            torch.cuda.synchronize()
            for got in (busy, other, again):
                assert torch.equal(_lib.check_ids(got.cpu()), quiet), rep


def test_softmax_selection_grouped_vs_literal():
    """Predictor.predict_batch's token rule, argmax(softmax(logits / T)) with sticky stop (predictor.py:295-297,333,343):
    the grouped kernel (arg max of the scaled logits) against the row-per-workgroup kernel (fp32 probabilities computed
    literally, then arg max) at BASELINE config 2's size, with and without temperature: same ids, incl. the -1 filler
    after a row's END."""
    d, cfg, sd_kw = load("primary_cfg2_clock")
    m, _ = model_for("primary_cfg2_clock", sd_kw, cfg)
    x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(DEV)
    with torch.no_grad():
        enc = m.encoder(x)
        for temp in (1.0, 0.7):
            grouped, _ = m.greedy_ids(enc, START, END, 150, temperature=temp, stop=_lib.STOP_STICKY, select=_lib.SELECT_SOFTMAX)
            literal, _ = m.greedy_ids(enc, START, END, 150, temperature=temp, stop=_lib.STOP_STICKY, select=_lib.SELECT_SOFTMAX,
                                      rows_per_workgroup=1)
            a, b = _lib.check_ids(grouped.cpu()).numpy(), literal.cpu().numpy()
            # free-running sequences may part at an fp32 near-tie of the two kernels' logits (different summation order)
            same_rows = (a == b).all(axis=1).mean()
            assert same_rows >= 0.97, (temp, same_rows)
            assert (a == -1).sum() > 0                       # some rows did stop early


def test_agent_scope_exchange_gives_identical_results():
    """The grouped kernels publish their exchange granules with workgroup-scope (L2-local) stores when the members of
    a group measure themselves on one XCD -- fast, but resting on gfx950's cache hierarchy rather than on the HSA memory
    model.  I2L_FLAG_AGENT_SCOPE_EXCHANGE keeps every store at agent scope (the conformant flavour): greedy ids, beam
    sequences / scores must be bit-identical to the default, one training step equal to rounding."""
    from img2latex_amd.training import TrainStep
    d, cfg, sd_kw = load("primary_cfg2_clock")
    m, _ = model_for("primary_cfg2_clock", sd_kw, cfg)
    x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(DEV)
    with torch.no_grad():
        enc = m.encoder(x)
        fast, _ = m.greedy_ids(enc, START, END, 150)
        fast8, _ = m.greedy_ids(enc, START, END, 150, flags=_lib.FLAG_DECODE_GROUP8)
        fast16, _ = m.greedy_ids(enc, START, END, 150, flags=_lib.FLAG_DECODE_GROUP16)
        fast_b, fast_s = m.beam_search_batch(enc[:37].contiguous(), START, END, 48, 5, return_scores=True)
        m.decoder.kernel_flags = _lib.FLAG_AGENT_SCOPE_EXCHANGE
        try:
            slow, _ = m.greedy_ids(enc, START, END, 150)
            slow8, _ = m.greedy_ids(enc, START, END, 150, flags=_lib.FLAG_DECODE_GROUP8)
            slow16, _ = m.greedy_ids(enc, START, END, 150, flags=_lib.FLAG_DECODE_GROUP16)
            assert m.decoder.group_status()["groups_on_one_xcd"] == 0           # the conformant flavour never claims the L2-local one
            slow_b, slow_s = m.beam_search_batch(enc[:37].contiguous(), START, END, 48, 5, return_scores=True)
        finally:
            m.decoder.kernel_flags = 0
    assert torch.equal(_lib.check_ids(slow.cpu()), _lib.check_ids(fast.cpu()))
    assert torch.equal(_lib.check_ids(slow8.cpu()), _lib.check_ids(fast8.cpu()))
    assert torch.equal(_lib.check_ids(slow16.cpu()), _lib.check_ids(fast16.cpu()))
    assert slow_b == fast_b and slow_s == fast_s
    cfg3 = synth.model_config(dropout=0.0)
    np_sd = synth.make_state_dict(cfg3, seed=42)
    xs = torch.from_numpy(synth.make_images(8, cfg3, seed=5)).to(DEV)
    forms = torch.from_numpy(synth.make_formulas(8, 40, cfg3["vocab_size"], seed=6)).to(DEV)
    outs = []
    for flags in (0, _lib.FLAG_AGENT_SCOPE_EXCHANGE):
        mm = Seq2SeqModel("cnn_lstm", cfg3["vocab_size"], synth.encoder_params(cfg3), synth.decoder_params(cfg3))
        mm.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in np_sd.items()})
        mm = mm.to(DEV)
        mm.decoder.kernel_flags = flags
        ts = TrainStep(mm)
        ts.step(xs, forms)
        outs.append(ts.flat_params.clone())
    # (the embedding-gradient scatter adds with atomics, so two runs of the SAME kernels agree to rounding, not to the bit)
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-6


def test_grouped_kernels_forced_timeout_fails_loudly_and_falls_back():
    """The failure path of the grouped kernels, forced: I2L_FLAG_TEST_DROP_MEMBER makes member 3 of every group exit at
    once, so the three others never see its placement granule; with I2L_FLAG_TEST_SHORT_TIMEOUT their polls give up
    after 2 ms (production: 10 ms + 50 us per step for a workgroup's first poll, 3 s afterwards).  Required: the launch
    RETURNS (no hang), greedy ids are -3 on every row and the logits NaN, beam lengths -3 -> the host entry points warn
    and produce the reference's result on the row-per-workgroup kernels; a training step's gradients are NaN and the
    fused clip + Adam skips the update."""
    import time
    from img2latex_amd.training import TrainStep
    d, cfg, sd_kw = load("primary_cfg2_clock")
    m, _ = model_for("primary_cfg2_clock", sd_kw, cfg)
    x = torch.from_numpy(synth.make_images(64, cfg, seed=1234)).to(DEV)
    bad = _lib.FLAG_TEST_DROP_MEMBER | _lib.FLAG_TEST_SHORT_TIMEOUT
    with torch.no_grad():
        enc = m.encoder(x)
        good, _ = m.greedy_ids(enc, START, END, 60)
        st = m.decoder.group_status()
        assert st == {"timed_out": False, "groups": 16, "groups_on_one_xcd": st["groups_on_one_xcd"]}
        record("grouped greedy decode, 16 groups: groups whose members share one XCD", st["groups_on_one_xcd"])
        tok0 = torch.full((64,), START, dtype=torch.int32, device=DEV)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids, logits, _ = m.decoder.run_steps(enc, 60, tok0, want_logits=True, flags=bad)
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 1.0                                  # milliseconds, not the 3 s limit
        assert bool((ids == -3).all()) and bool(torch.isnan(logits[:, :, 0]).all())
        assert m.decoder.group_status()["timed_out"]
        with pytest.raises(RuntimeError):
            _lib.check_ids(ids.cpu())
        m.decoder.kernel_flags = bad
        try:
            with pytest.warns(RuntimeWarning):
                via_host = m.greedy_ids_host(enc, START, END, 60)              # falls back to the row-per-workgroup kernel
            with pytest.warns(UserWarning):
                beams = m.beam_search_batch(enc[:10].contiguous(), START, END, 40, 5)
        finally:
            m.decoder.kernel_flags = 0
        # the fallback kernel sums in another order than the grouped one: judged against the REFERENCE's ids (margin guard)
        ref_ids = d["ids"].astype(np.int64)[:64]
        n = min(60, ref_ids.shape[1] - 1)
        for got_ids in (via_host, good.cpu()):
            assert _margin_guard(got_ids.numpy()[:, :n], ref_ids, d["margins"][:64], tol=2e-4) <= 3
        assert beams == m.beam_search_batch(enc[:10].contiguous(), START, END, 40, 5, flags=_lib.FLAG_NO_GROUP)
    cfg3 = synth.model_config(dropout=0.0)
    mm = Seq2SeqModel("cnn_lstm", cfg3["vocab_size"], synth.encoder_params(cfg3), synth.decoder_params(cfg3))
    mm.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg3, seed=42).items()})
    mm = mm.to(DEV)
    ts = TrainStep(mm)
    before = ts.flat_params.clone()
    xs = torch.from_numpy(synth.make_images(8, cfg3, seed=5)).to(DEV)
    forms = torch.from_numpy(synth.make_formulas(8, 40, cfg3["vocab_size"], seed=6)).to(DEV)
    mm.decoder.kernel_flags = bad
    out = ts.step(xs, forms)
    assert float(out["skipped"]) == 1.0 and torch.equal(ts.flat_params, before)
    mm.decoder.kernel_flags = 0
    out = ts.step(xs, forms)
    assert float(out["skipped"]) == 0.0 and np.isfinite(float(out["loss"]))


def test_encoder_reuses_packed_filters_until_a_weight_changes():
    """Inference keeps the packed (3 x bf16) filter images of conv blocks 2 / 3 from batch to batch
    (I2L_FLAG_WEIGHTS_PACKED): a second batch gives what a fresh model gives, and an in-place weight update (version
    bump) is picked up by the next call."""
    d, cfg, sd_kw = load("primary")
    m, _ = model_for("primary")
    fresh = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
    fresh.load_state_dict(m.state_dict())
    fresh = fresh.to(DEV).eval()
    xa, xb = images(cfg, 5, seed=1, device=DEV), images(cfg, 5, seed=2, device=DEV)
    with torch.no_grad():
        m.encoder(xa)
        got = m.encoder(xb)                                  # second call: packed filters reused
        want = fresh.encoder(xb)                             # first call of another object: packs
        assert torch.equal(got, want)
        w = m.encoder.cnn_layers[3].weight
        w.mul_(1.25)                                         # in place: same storage, new version
        fresh.encoder.cnn_layers[3].weight.copy_(w)
        fresh2 = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
        fresh2.load_state_dict(fresh.state_dict())
        fresh2 = fresh2.to(DEV).eval()
        assert torch.equal(m.encoder(xb), fresh2.encoder(xb))
        w.div_(1.25)


@pytest.mark.parametrize("members", [8, 16])
@pytest.mark.parametrize("fname", ["primary_cfg2_clock", "primary_cfg2"])
def test_group8_decode_ids_vs_reference(fname, members):
    """decode_group8_kernel (8 members x 8 rows per group: one wave per SIMD, ~80 KB of LDS, the footprint that shares a
    CU with a conv workgroup) and decode_group16_kernel (r04: 16 members x 16 rows, the per-step products on the matrix
    cores as split-bf16 MFMA bursts) on BASELINE configs[1] at full size: ids against the reference's fixture with the
    margin guard, both stop rules, temperature, ragged batch sizes (last group partly empty), the forced time-out path."""
    d, cfg, sd_kw = load(fname)
    m, _ = model_for(fname, sd_kw, cfg)
    ref_ids = d["ids"].astype(np.int64)
    steps = ref_ids.shape[1] - 1
    x = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(DEV)
    G8 = _lib.FLAG_DECODE_GROUP8 if members == 8 else _lib.FLAG_DECODE_GROUP16
    with torch.no_grad():
        enc = m.encoder(x)
        ids8, _ = m.greedy_ids(enc, START, END, 150, flags=G8)
        st = m.decoder.group_status()
        assert st["groups"] == 256 // members and not st["timed_out"]       # 256 rows / 8 (16): it WAS that kernel
        record(f"{fname} group{members} decode, {256 // members} groups: groups whose {members} members share one XCD", st["groups_on_one_xcd"])
        ids4, _ = m.greedy_ids(enc, START, END, 150)
        got8 = _lib.check_ids(ids8.cpu()).numpy()[:, :steps]
        assert _margin_guard(got8, ref_ids, d["margins"], tol=2e-4) <= 0.05 * 256
        assert int((got8 != _lib.check_ids(ids4.cpu()).numpy()[:, :steps]).any(axis=1).sum()) <= 3   # near-ties only
        # sticky stop + argmax of softmax, temperature
        a, _ = m.greedy_ids(enc, START, END, 60, temperature=0.7, stop=_lib.STOP_STICKY, select=_lib.SELECT_SOFTMAX, flags=G8)
        b, _ = m.greedy_ids(enc, START, END, 60, temperature=0.7, stop=_lib.STOP_STICKY, select=_lib.SELECT_SOFTMAX)
        a, b = _lib.check_ids(a.cpu()), _lib.check_ids(b.cpu())
        assert int((a != b).any(dim=1).sum()) <= 3
        # ragged batches: 1 .. 33 rows, 77 rows (last group holds 5 / 13)
        for B in (1, 7, 8, 9, 15, 16, 17, 19, 33, 77):
            e = enc[:B].contiguous()
            p8, _ = m.greedy_ids(e, START, END, 40, flags=G8)
            p4, _ = m.greedy_ids(e, START, END, 40)
            p8, p4 = _lib.check_ids(p8.cpu()).numpy(), _lib.check_ids(p4.cpu()).numpy()
            assert _margin_guard(p8, ref_ids[:B], d["margins"][:B], tol=2e-4) <= 1, B
            assert p8.shape == p4.shape
        # forced time-out: returns in milliseconds with ids -3
        tok0 = torch.full((64,), START, dtype=torch.int32, device=DEV)
        bad, _, _ = m.decoder.run_steps(enc[:64].contiguous(), 40, tok0,
                                        flags=G8 | _lib.FLAG_TEST_DROP_MEMBER | _lib.FLAG_TEST_SHORT_TIMEOUT)
        assert bool((bad == -3).all())


def test_coresident_pipeline_ids_vs_reference():
    """GreedyPipeline with the 8-member grouped decode (FLAG_DECODE_GROUP8) sharing every CU with the next batch's conv
    workgroups: six batches of BASELINE configs[1] (three distinct image sets, so that consecutive batches differ) through
    the two-stream pipeline -- every batch's ids against the reference's fixture (margin guard) for the fixture's images
    and equal to the one-batch-at-a-time search for the others up to near-ties; nothing times out."""
    from img2latex_amd.pipeline import GreedyPipeline
    d, cfg, sd_kw = load("primary_cfg2")
    m, _ = model_for("primary_cfg2", sd_kw, cfg)
    ref_ids = d["ids"].astype(np.int64)
    steps = ref_ids.shape[1] - 1
    sets = [torch.from_numpy(synth.make_images(256, cfg, seed=s)).to(DEV) for s in (1234, 77, 78)]
    with torch.no_grad():
        want = [_lib.check_ids(m.greedy_ids(m.encoder(x), START, END, 150)[0].cpu()).numpy() for x in sets]
    pipe = GreedyPipeline(m, START, END, 150, depth=2, rows_per_workgroup=0, decode_flags=_lib.FLAG_DECODE_GROUP8,
                          decode_priority=-1)
    order = [0, 1, 2, 0, 2, 1]
    got = []
    for i in order:
        if pipe.pending() >= 2:
            got.append(pipe.collect().numpy().copy())
        pipe.submit(sets[i])
    while pipe.pending():
        got.append(pipe.collect().numpy().copy())
    assert len(got) == 6
    for i, g in zip(order, got):
        assert int((g != want[i]).any(axis=1).sum()) <= 3, i               # the other kernel's sums: near-ties only
        if i == 0:
            assert _margin_guard(g[:, :steps], ref_ids, d["margins"], tol=2e-4) <= 0.05 * 256


def test_pipeline_with_two_encoder_streams_resnet():
    """GreedyPipeline(encoder_streams=2): two ResNet trunks in flight beside the decode (bench.py --mode resnet).  Seven
    batches of three distinct image sets through it: every batch's ids equal the one-batch-at-a-time search on the same
    kernels' encoder output (the trunk is deterministic per batch, the 8-member decode may differ from the 4-member one only
    at fp32 near-ties), results come back in submission order, per-stream workspaces do not collide."""
    from img2latex_amd.model import Seq2SeqModel
    from img2latex_amd.pipeline import GreedyPipeline
    cfg = synth.model_config()
    enc_p = dict(img_height=64, img_width=320, channels=3, model_name="resnet18", embedding_dim=256, freeze_backbone=True)
    m = Seq2SeqModel("resnet_lstm", cfg["vocab_size"], enc_p, synth.decoder_params(cfg))
    shapes = [(k, tuple(v.shape)) for k, v in m.encoder.state_dict().items()]
    full = {"encoder." + k: torch.from_numpy(v) for k, v in synth.make_resnet_state_dict(shapes, seed=5).items()}
    full.update({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0).items()
                 if k.startswith("decoder.")})
    m.load_state_dict(full)
    m = m.to(DEV).eval()
    sets = [torch.from_numpy(synth.make_images(96, cfg, seed=s)).to(DEV) for s in (11, 12, 13)]
    pipe = GreedyPipeline(m, START, END, 40, rows_per_workgroup=0, decode_flags=_lib.FLAG_DECODE_GROUP8, encoder_streams=2)
    assert pipe.depth >= 3
    # with two trunks in flight the pipeline asks for the throughput shapes of the trunk kernels (128-column ring tiles: another
    # fp32 summation order than the balanced shapes a lone trunk takes, r04): the reference pass runs the same kernels
    assert pipe.encoder_flags == (_lib.FLAG_RESNET_WIDE_TILES | _lib.FLAG_RESNET_NO_PATCH)
    with torch.no_grad():
        m.encoder.kernel_flags = pipe.encoder_flags
        encs = [m.encoder(x) for x in sets]
        m.encoder.kernel_flags = 0
        want = [_lib.check_ids(m.greedy_ids(e, START, END, 40, flags=_lib.FLAG_DECODE_GROUP8)[0].cpu()).numpy() for e in encs]
        assert not np.array_equal(want[0], want[1])
    order = [0, 1, 2, 2, 0, 1, 0]
    got = []
    for i in order:
        if pipe.pending() >= pipe.depth:
            got.append(pipe.collect().numpy().copy())
        pipe.submit(sets[i])
    while pipe.pending():
        got.append(pipe.collect().numpy().copy())
    assert len(got) == len(order)
    for i, g in zip(order, got):
        assert np.array_equal(g, want[i]), i


def test_pipeline_with_two_encoder_streams_cnn():
    """GreedyPipeline(encoder_streams=2) over the CNN encoder: two forwards of ONE CNNEncoder run side by
    side, so its FC split-K slabs and packed filter images must be per stream.  Eight batches of three distinct image sets:
    every batch's ids equal the one-batch-at-a-time search on the same decode kernel, in submission order."""
    from img2latex_amd.pipeline import GreedyPipeline
    d, cfg, sd_kw = load("primary_cfg2")
    m, _ = model_for("primary_cfg2", sd_kw, cfg)
    sets = [torch.from_numpy(synth.make_images(256, cfg, seed=s)).to(DEV) for s in (1234, 77, 78)]
    with torch.no_grad():
        want = [_lib.check_ids(m.greedy_ids(m.encoder(x), START, END, 60, flags=_lib.FLAG_DECODE_GROUP8)[0].cpu()).numpy()
                for x in sets]
    assert not np.array_equal(want[0], want[1])
    pipe = GreedyPipeline(m, START, END, 60, rows_per_workgroup=0, decode_flags=_lib.FLAG_DECODE_GROUP8, encoder_streams=2)
    order = [0, 1, 2, 2, 0, 1, 0, 2]
    got = []
    for i in order:
        if pipe.pending() >= pipe.depth:
            got.append(pipe.collect().numpy().copy())
        pipe.submit(sets[i])
    while pipe.pending():
        got.append(pipe.collect().numpy().copy())
    assert len(got) == len(order)
    for i, g in zip(order, got):
        assert np.array_equal(g, want[i]), i
    assert len({k[1] for k in m.encoder._packed_ws}) >= 2               # one packed image per (block, stream)
    n_slots = len([k for k in m.decoder._ws_by_stream if isinstance(k, tuple) and k[0] == "slot"])
    assert n_slots >= 1
    pipe.close()                                                        # the pipeline's decoder workspaces go back
    assert not [k for k in m.decoder._ws_by_stream if isinstance(k, tuple) and k[0] == "slot"]


def test_residency_signal_and_bounded_stream_wait():
    """i2l_greedy_decode_ex's residency signal + i2l_stream_wait_value32 (r04: GreedyPipeline's dependency between decode(i)
    and encoder(i + 1), where r03 had a 30 us delay kernel), through the C ABI:
    the grouped kernels (4 and 8 members) publish the value once their groups are resident, an ungrouped launch publishes at
    once, a launch with a silent member (forced time-out) never does; a wait on a published value returns at once, a wait
    on a value that never comes returns after its bound; and in a traced pipeline every encoder starts after the decode
    launch it was held for, with identical ids."""
    import ctypes
    import time
    from img2latex_amd.pipeline import GreedyPipeline
    d, cfg, sd_kw = load("primary_cfg2_clock")
    m, _ = model_for("primary_cfg2_clock", sd_kw, cfg)
    L = _lib.lib()
    x = torch.from_numpy(synth.make_images(64, cfg, seed=1234)).to(DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    with torch.no_grad():
        enc = m.encoder(x)
        base = m.greedy_ids(enc, START, END, 40)[0].cpu()
        for n, (fl, rows_per_wg) in enumerate([(0, 0), (_lib.FLAG_DECODE_GROUP8, 0), (0, 1)], start=1):
            ids, _, _ = m.decoder.run_steps(enc, 40, torch.full((64,), START, dtype=torch.int32, device=DEV), flags=fl,
                                            rows_per_workgroup=rows_per_wg, resident=(flag, 100 + n))
            torch.cuda.synchronize()
            assert int(flag.item()) == 100 + n, (fl, rows_per_wg)
            assert int((ids.cpu() != base).any(dim=1).sum()) <= 2           # the signal does not touch the results
        bad = _lib.FLAG_DECODE_GROUP8 | _lib.FLAG_TEST_DROP_MEMBER | _lib.FLAG_TEST_SHORT_TIMEOUT
        ids, _, _ = m.decoder.run_steps(enc, 40, torch.full((64,), START, dtype=torch.int32, device=DEV), flags=bad,
                                        resident=(flag, 999))
        torch.cuda.synchronize()
        assert bool((ids == -3).all()) and int(flag.item()) == 103           # a group that never assembled says nothing
    # the wait: satisfied (wrapping compare: 103 - 50 >= 0), then a value that never comes
    s = torch.cuda.Stream()
    assert L.i2l_stream_wait_value32(flag.data_ptr(), 50, 5000.0, s.cuda_stream) == 0
    s.synchronize()
    t0 = time.perf_counter()
    assert L.i2l_stream_wait_value32(flag.data_ptr(), 104, 3000.0, s.cuda_stream) == 0
    s.synchronize()
    waited = time.perf_counter() - t0
    assert 2.5e-3 <= waited <= 0.5, waited
    assert L.i2l_stream_wait_value32(None, 1, 10.0, s.cuda_stream) == -1 and \
        L.i2l_stream_wait_value32(flag.data_ptr(), 1, 1e9, s.cuda_stream) == -1
    # the pipeline holds every encoder behind the decode launch it depends on
    x256 = torch.from_numpy(synth.make_images(256, cfg, seed=1234)).to(DEV)
    for hold in (True, False):
        pipe = GreedyPipeline(m, START, END, 150, rows_per_workgroup=0, decode_flags=_lib.FLAG_DECODE_GROUP8, decode_priority=-1,
                              hold_encoder=hold)
        pipe.trace = []
        outs = []
        for _ in range(12):
            if pipe.pending() >= pipe.depth:
                outs.append(pipe.collect().clone())
            pipe.submit(x256)
        while pipe.pending():
            outs.append(pipe.collect().clone())
        tr = pipe.trace
        offs = [a["dec_start"].elapsed_time(b["enc_start"]) for a, b in zip(tr[:-1], tr[1:])]
        record(f"pipeline hold_encoder={hold}: fraction of encoders that started after the previous decode's launch",
               float(np.mean([o > 0 for o in offs])))
        if hold:
            assert all(o > 0 for o in offs), offs
            assert int(pipe._resident.item()) == 12
        assert all(torch.equal(o, outs[0]) for o in outs)
        pipe.close()


def test_pipeline_falls_back_when_the_grouped_decode_times_out():
    """A batch whose grouped decode timed out (forced: silent member + 2 ms limits) is decoded again on the
    row-per-workgroup kernel inside GreedyPipeline.collect(): same ids as the healthy pipeline, one warning per batch."""
    from img2latex_amd.pipeline import GreedyPipeline
    d, cfg, sd_kw = load("primary_cfg2_clock")
    m, _ = model_for("primary_cfg2_clock", sd_kw, cfg)
    x = torch.from_numpy(synth.make_images(64, cfg, seed=1234)).to(DEV)
    good = GreedyPipeline(m, START, END, 60, rows_per_workgroup=0, decode_flags=_lib.FLAG_DECODE_GROUP8)
    good.submit(x)
    want = good.collect().clone()
    bad = GreedyPipeline(m, START, END, 60, rows_per_workgroup=0,
                         decode_flags=_lib.FLAG_DECODE_GROUP8 | _lib.FLAG_TEST_DROP_MEMBER | _lib.FLAG_TEST_SHORT_TIMEOUT)
    bad.submit(x)
    bad.submit(x)
    with pytest.warns(RuntimeWarning):
        a = bad.collect().clone()
    with pytest.warns(RuntimeWarning):
        b = bad.collect().clone()
    assert torch.equal(a, b)
    assert int((a != want).any(dim=1).sum()) <= 2                      # the fallback kernel's sums: near-ties only
