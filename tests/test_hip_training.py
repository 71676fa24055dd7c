"""GPU parity of the training path against the oracle's autograd (and the golden G6 fixtures)."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import img2latex_oracle as O
from conftest import record
from helpers import (ALL, BIG, PAD, SMALL, adam_first_step_allowance, check_decisions, hip_decisions, images, load, sample,
                     torch_state_dict)
from img2latex_amd import _lib, synth
from img2latex_amd.model import Seq2SeqModel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_close(a, b, tol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-12)
    err = float(np.abs(a - b).max())
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def abs_close(a, b, tol, what):
    """north_star: logits within 1e-4 (absolute); the measured maximum is recorded."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float(np.abs(a - b).max())
    record(what + " [abs]", err)
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol}"


def build(name):
    d, cfg, sd_kw = load(name)
    m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, **sd_kw).items()})
    return d, cfg, m.to(DEV)


@pytest.mark.parametrize("name,batch", [(n, 4) for n in SMALL + ["primary"]] + [("primary", 7), ("primary", 1), ("primary", 70), ("primary", 130)])
def test_decoder_backward_vs_oracle(name, batch):
    """d(loss)/d(decoder params) and d(loss)/d(enc) for the teacher-forced decoder, dropout 0.  The primary config
    runs the grouped recurrences: 4 workgroups per ONE row up to 64 rows (batches 1, 4, 7), per TWO rows up to 128 (batch
    70), per FOUR rows above (batch 130: 33 groups, the last with two missing rows)."""
    d, cfg, m = build(name)
    sd = torch_state_dict(name)
    T = 12
    forms = torch.from_numpy(synth.make_formulas(batch, T, cfg["vocab_size"], seed=777, min_len=5))
    x = images(cfg) if batch == 4 else torch.from_numpy(synth.make_images(batch, cfg, seed=99))
    with torch.no_grad():
        enc_cpu = O.cnn_encoder(sd, cfg, x)
    # oracle: autograd through the restated decoder + CE
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("decoder.")}
    enc_leaf = enc_cpu.clone().requires_grad_(True)
    logits_cpu = O.decoder_forward({**sd, **params}, cfg, enc_leaf, forms[:, :-1])
    loss_cpu = O.ce_label_smooth(logits_cpu, forms[:, 1:], PAD)
    loss_cpu.backward()
    # HIP: the same through torch autograd + our Function, torch's own CE on the device
    m.train()
    enc_dev = enc_cpu.to(DEV).requires_grad_(True)
    logits = m.decoder(enc_dev, forms[:, :-1].to(DEV))
    abs_close(logits.detach().cpu().numpy(), logits_cpu.detach().numpy(), 1e-4, f"{name} B={batch} training-forward logits")
    crit = torch.nn.CrossEntropyLoss(ignore_index=PAD, reduction="mean", label_smoothing=0.1)
    loss = crit(logits.transpose(1, 2), forms[:, 1:].to(DEV))
    loss.backward()
    assert abs(float(loss) - float(loss_cpu)) < 1e-5 * max(1.0, abs(float(loss_cpu)))
    rel_close(enc_dev.grad.cpu().numpy(), enc_leaf.grad.numpy(), 2e-4, "d enc")
    for n, p in m.decoder.named_parameters():
        ref = params["decoder." + n].grad
        if n.startswith("attention."):
            assert float(p.grad.abs().max()) == 0.0 and float(ref.abs().max()) == 0.0
            continue
        rel_close(p.grad.cpu().numpy(), ref.numpy(), 2e-4, n)


def test_ce_kernel_vs_torch():
    torch.manual_seed(0)
    L = _lib.lib()
    for rows, V in [(7, 37), (48, 512), (301, 50)]:
        logits = torch.randn(rows, V) * 3
        tgt = torch.randint(0, V, (rows,))
        tgt[::5] = PAD
        ref_logits = logits.clone().requires_grad_(True)
        loss = torch.nn.functional.cross_entropy(ref_logits, tgt, ignore_index=PAD, reduction="sum", label_smoothing=0.1)
        loss.backward()
        ld, td = logits.to(DEV), tgt.to(torch.int32).to(DEV)
        dl = torch.empty_like(ld)
        out = torch.empty(2, device=DEV)
        nbytes = L.i2l_ce_workspace_bytes(rows)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        assert L.i2l_ce_label_smooth_fwd_bwd(ld.data_ptr(), td.data_ptr(), rows, V, PAD, 0.1, ws.data_ptr(), nbytes,
                                             dl.data_ptr(), out.data_ptr(), _lib.stream_ptr()) == 0
        got = out.cpu()
        assert abs(float(got[0]) - float(loss)) <= 1e-5 * abs(float(loss))
        assert int(got[1]) == int((tgt != PAD).sum())
        rel_close(dl.cpu().numpy(), ref_logits.grad.numpy(), 1e-5, "dlogits")


@pytest.mark.parametrize("variant", ["default", "exact"])
@pytest.mark.parametrize("name", SMALL + ["primary"])
def test_encoder_backward_vs_oracle(name, variant):
    """d(sum(enc * r))/d(encoder params) against the oracle's autograd.
    The conv gradients are sums of ~10^5..10^6 products of both signs that cancel to ~1e-3 of their magnitude, and they
    are discontinuous in the forward pass's discrete choices (pooling arg max, ReLU gate): two correct fp32 evaluations
    -- the oracle in fp32 vs the oracle in fp64 -- differ by up to 4e-3 of a gradient's maximum wherever one window is
    a near-tie (recorded).  So: (1) the choices of the HIP forward may differ from float64's only at near-ties fp32
    cannot resolve (helpers.check_decisions); (2) element by element the gradients are compared with the FLOAT64
    gradient under those same choices -- smooth rounding only, 5e-5 of each gradient's maximum; (3) every gradient's
    norm is within 3e-4 of the fp32 oracle's.  `default` runs the forward on the 3 x bf16 split kernels, whose near-tie
    windows are re-evaluated with fp32 FMAs (conv_pool_fixup_kernel); `exact` is I2L_FLAG_EXACT_FP32."""
    d, cfg, m = build(name)
    sd = torch_state_dict(name)
    x = images(cfg)
    r = torch.from_numpy(synth.uniform(5, "probe", (4, cfg["embedding_dim"]), -1.0, 1.0))

    def oracle(dtype, decisions=None):
        params = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd.items() if k.startswith("encoder.")}
        enc_ = O.cnn_encoder({**{k: v.to(dtype) for k, v in sd.items()}, **params}, cfg, x.to(dtype), decisions)
        (enc_ * r.to(dtype)).sum().backward()
        return enc_.detach(), {k: p.grad for k, p in params.items()}
    enc_cpu, g32 = oracle(torch.float32)
    _, g64 = oracle(torch.float64)
    m.train()
    m.encoder.kernel_flags = {"default": 0, "exact": _lib.FLAG_EXACT_FP32}[variant]
    decisions = hip_decisions(m, x.to(DEV))
    n_off, n_all = check_decisions(sd, cfg, x, decisions)
    record(f"{name} B=4 HIP {variant}: windows deciding differently from float64 [of {n_all}]", n_off)
    _, g64d = oracle(torch.float64, decisions)
    enc = m.encoder(x.to(DEV))
    rel_close(enc.detach().cpu().numpy(), enc_cpu.numpy(), 1e-5, "enc")
    (enc * r.to(DEV)).sum().backward()
    for n, p in m.encoder.named_parameters():
        t = g64d["encoder." + n]
        scale = float(t.abs().max())
        g = p.grad.cpu().double()
        e_hip = float((g - t).abs().max()) / scale
        e_ref = float((g32["encoder." + n].double() - g64["encoder." + n]).abs().max()) / float(g64["encoder." + n].abs().max())
        record(f"{name} B=4 d{n} vs fp64 oracle under the same decisions [rel to max]: HIP {variant}", e_hip)
        record(f"{name} B=4 d{n}: fp32 oracle vs fp64 oracle, each with its own decisions [rel to max]", e_ref)
        assert e_hip <= 5e-5, (n, e_hip)
        n32 = float(g32["encoder." + n].norm())
        assert abs(float(g.norm()) - n32) <= 3e-4 * n32, n


def test_training_forward_decisions_match_fp32():
    """The pooling arg max and the ReLU gate of the split-product training forward against a float64 evaluation, on
    the three block shapes of the primary config at 16 images: every window where the two disagree must be a genuine
    fp32 near-tie (top-two gap / pre-activation below 2e-6 of the window's magnitude -- where fp32 evaluations disagree
    among themselves), and the exact-fp32 kernels must not do better by more than a handful of windows.  Data with
    exact ties (constant regions) must give the first index in both."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)

    def run(x, w, b, flags):
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        y = torch.empty(B, Cout, H // 2, W // 2, device=DEV)
        am = torch.empty(B, Cout, H // 2, W // 2, dtype=torch.uint8, device=DEV)
        nbytes = L.i2l_conv_workspace_bytes(Cin, Cout)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=DEV)
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)                   # named: the device copies must outlive the launch
        assert L.i2l_conv3x3_relu_pool2_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(),
                                            am.data_ptr(), B, Cin, H, W, Cout, ws.data_ptr(), nbytes, flags,
                                            _lib.stream_ptr()) == 0
        return y.cpu(), am.cpu()

    for (Cin, Cout, H, W) in [(3, 32, 64, 320), (32, 64, 32, 160), (64, 128, 16, 80)]:
        x = torch.rand(16, Cin, H, W, generator=g) * 2 - 1
        if Cin == 3:
            x[:, :, :, 200:] = 1.0                                     # white margin: exact ties
        w = (torch.rand(Cout, Cin, 3, 3, generator=g) * 2 - 1) / (Cin * 9) ** 0.5
        b = (torch.rand(Cout, generator=g) * 2 - 1) * 0.1
        conv = F.conv2d(x.double(), w.double(), None, padding=1)       # float64 pre-bias
        B_, C_, Hc, Wc = conv.shape
        quads = conv.reshape(B_, C_, Hc // 2, 2, Wc // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(B_, C_, Hc // 2, Wc // 2, 4)
        top2 = quads.topk(2, dim=-1).values
        ref_am = quads.argmax(dim=-1)                                  # first maximum
        gap = top2[..., 0] - top2[..., 1]
        pre = top2[..., 0] + b.double()[None, :, None, None]
        mag = quads.abs().amax(dim=-1).clamp_min(1e-30)
        counts = {}
        for flags in (0, _lib.FLAG_EXACT_FP32):
            y, am = run(x, w, b, flags)
            live = pre > 0                                             # the arg max only matters where the ReLU passes
            wrong_am = (am.long() != ref_am) & live & (gap > 0)
            wrong_gate = (y > 0) != (pre > 0)
            assert float((gap[wrong_am] / mag[wrong_am]).max() if wrong_am.any() else 0.0) <= 2e-6, (Cin, flags)
            assert float((pre[wrong_gate].abs() / mag[wrong_gate]).max() if wrong_gate.any() else 0.0) <= 2e-6, (Cin, flags)
            tie = (gap == 0) & live
            assert bool((am.long()[tie] == ref_am[tie]).all()), "exact ties take the first index"
            counts[flags] = int(wrong_am.sum()) + int(wrong_gate.sum())
            record(f"training forward decisions Cin={Cin} flags={flags} [windows off fp64 of {ref_am.numel()}]", counts[flags])
        assert counts[0] <= counts[_lib.FLAG_EXACT_FP32] + 8 + ref_am.numel() // 500000, counts


def test_training_forward_near_tie_list_overflow():
    """An input made of near-ties (a constant image plus 1e-6 noise: almost every pooling window's runner-up is within
    2^-13 of its maximum) overflows the fix-up list (1 M entries; 1.3 M windows here).  Nothing may break: the call
    succeeds, the pooled values are the fp32-grade maxima, and every arg max points at a value within 1e-5 of its
    window's maximum -- listed windows are re-evaluated in fp32, the overflow keeps the split-product decision."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    B, Cin, H, W, Cout = 8, 3, 64, 320, 32
    x = 0.5 + 1e-6 * torch.randn(B, Cin, H, W, generator=g)
    w = (torch.rand(Cout, Cin, 3, 3, generator=g) * 2 - 1) / 27 ** 0.5
    b = torch.rand(Cout, generator=g) * 0.1 + 0.5                      # keeps the ReLU open on all but one channel
    y = torch.empty(B, Cout, H // 2, W // 2, device=DEV)
    am = torch.empty(B, Cout, H // 2, W // 2, dtype=torch.uint8, device=DEV)
    nbytes = L.i2l_conv_workspace_bytes(Cin, Cout)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)                       # named: the device copies must outlive the launch
    assert L.i2l_conv3x3_relu_pool2_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(),
                                        am.data_ptr(), B, Cin, H, W, Cout, ws.data_ptr(), nbytes, 0, _lib.stream_ptr()) == 0
    listed = int(ws[:4].view(torch.int32).cpu()[0])
    assert listed > (1 << 20), listed                                  # the list did overflow
    win = O.pool_windows(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    best = win.amax(-1)
    rel_close(y.cpu().numpy(), torch.relu(best).float().numpy(), 1e-5, "pooled values")
    chosen = win.gather(-1, am.cpu().long().unsqueeze(-1)).squeeze(-1)
    live = best > 1e-2                                                 # where the ReLU passes (one channel cancels to ~-8e-4: gate closed)
    assert int(live.sum()) > 0.9 * live.numel()
    assert float(((best - chosen)[live] / best[live]).max()) <= 1e-5


def test_conv_bwd_odd_shapes():
    torch.manual_seed(2)
    L = _lib.lib()
    for (B, Cin, H, W, Cout) in [(2, 3, 7, 9, 5), (3, 5, 16, 70, 9), (2, 8, 12, 20, 32), (1, 32, 16, 48, 64), (2, 33, 9, 11, 96),
                                 # the dedicated weight-gradient kernel (Cin % 32 == 0, Cout % 64 == 0, W % 16 == 0): one and
                                 # four channel-block combinations, strips shorter / longer than a workgroup's run, H = 2
                                 (4, 32, 32, 160, 64), (3, 64, 16, 80, 128), (16, 32, 10, 32, 64), (40, 64, 2, 64, 64)]:
        x = torch.randn(B, Cin, H, W, requires_grad=True)
        w = (torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)).requires_grad_(True)
        b = torch.randn(Cout, requires_grad=True)
        y = O.conv_block(x, w, b)
        dy = torch.randn_like(y)
        (y * dy).sum().backward()
        xd, wd, bd, dyd = x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV), dy.to(DEV)
        yd = torch.empty(y.shape, device=DEV)
        am = torch.empty(y.shape, dtype=torch.uint8, device=DEV)
        nb = L.i2l_conv_workspace_bytes(Cin, Cout)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
        assert L.i2l_conv3x3_relu_pool2_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), am.data_ptr(), B,
                                            Cin, H, W, Cout, ws.data_ptr(), nb, 0, _lib.stream_ptr()) == 0
        dx, dw, db = torch.empty_like(xd), torch.empty_like(wd), torch.empty_like(bd)
        nb2 = L.i2l_conv_bwd_workspace_bytes(B, Cin, H, W, Cout)
        ws2 = torch.empty(nb2, dtype=torch.uint8, device=DEV)
        assert L.i2l_conv3x3_relu_pool2_bwd(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), am.data_ptr(), dyd.data_ptr(),
                                            dx.data_ptr(), dw.data_ptr(), db.data_ptr(), B, Cin, H, W, Cout,
                                            ws2.data_ptr(), nb2, 0, None, _lib.stream_ptr()) == 0
        rel_close(dx.cpu().numpy(), x.grad.numpy(), 2e-4, "dx")
        rel_close(dw.cpu().numpy(), w.grad.numpy(), 2e-4, "dw")
        rel_close(db.cpu().numpy(), b.grad.numpy(), 2e-4, "db")


def test_first_block_sparse_weight_gradient():
    """dx == NULL, Cin <= 3, Cout % 32 == 0: conv_wgrad_first_kernel (one product per pooled cell, channel and tap, exact fp32)
    against autograd in float64 and against the implicit-im2col GEMM it replaces (FLAG_CONV_NO_SPARSE_WGRAD); odd heights /
    widths (the last row / column is outside every pooling window), one and two channel blocks, a band that ends past the
    last pooled row."""
    torch.manual_seed(4)
    L = _lib.lib()
    for (B, Cin, H, W, Cout) in [(3, 3, 64, 320, 32), (2, 1, 16, 36, 32), (2, 3, 17, 33, 64), (5, 2, 6, 10, 32),
                                 (1, 3, 2, 2, 32), (2, 1, 64, 800, 32)]:
        x = torch.randn(B, Cin, H, W, dtype=torch.float64)
        w = (torch.randn(Cout, Cin, 3, 3, dtype=torch.float64) / (3 * Cin ** 0.5)).requires_grad_(True)
        b = torch.randn(Cout, dtype=torch.float64, requires_grad=True)
        xd, wd, bd = x.float().to(DEV), w.detach().float().to(DEV), b.detach().float().to(DEV)
        yd = torch.empty((B, Cout, H // 2, W // 2), device=DEV)
        am = torch.empty(yd.shape, dtype=torch.uint8, device=DEV)
        nb = L.i2l_conv_workspace_bytes(Cin, Cout)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
        assert L.i2l_conv3x3_relu_pool2_fwd(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), yd.data_ptr(), am.data_ptr(), B,
                                            Cin, H, W, Cout, ws.data_ptr(), nb, 0, _lib.stream_ptr()) == 0
        dy = torch.randn(yd.shape, dtype=torch.float64)
        # float64 autograd UNDER THE HIP FORWARD'S decisions (which window position won, which outputs are positive)
        z = torch.nn.functional.conv2d(x.float().double(), w, b, padding=1)
        Hp, Wp = H // 2, W // 2
        zc = z[:, :, :2 * Hp, :2 * Wp].reshape(B, Cout, Hp, 2, Wp, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, Cout, Hp, Wp, 4)
        pick = torch.gather(zc, 4, am.cpu().long().unsqueeze(-1)).squeeze(-1)
        mask = (yd.cpu() > 0).double()
        (pick * mask * dy).sum().backward()
        dyd = dy.float().to(DEV)
        nb2 = L.i2l_conv_bwd_workspace_bytes(B, Cin, H, W, Cout)
        outs = []
        for fl in (0, _lib.FLAG_CONV_NO_SPARSE_WGRAD):
            dw = torch.full_like(wd, float("nan"))
            db = torch.full_like(bd, float("nan"))
            ws2 = torch.empty(nb2, dtype=torch.uint8, device=DEV)
            assert L.i2l_conv3x3_relu_pool2_bwd(xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), am.data_ptr(), dyd.data_ptr(),
                                                None, dw.data_ptr(), db.data_ptr(), B, Cin, H, W, Cout,
                                                ws2.data_ptr(), nb2, fl, None, _lib.stream_ptr()) == 0
            outs.append((dw.cpu().numpy(), db.cpu().numpy()))
        tag = f"first-block wgrad {B}x{Cin}x{H}x{W}->{Cout}"
        rel_close(outs[0][0], w.grad.numpy(), 2e-5, tag + " dw (sparse kernel vs float64)")
        rel_close(outs[0][1], b.grad.numpy(), 2e-5, tag + " db (sparse kernel vs float64)")
        rel_close(outs[1][0], w.grad.numpy(), 2e-4, tag + " dw (GEMM vs float64)")
        rel_close(outs[0][0], outs[1][0], 2e-4, tag + " dw (sparse kernel vs GEMM)")


def _params_after_vs_fixture(name, cfg, names, sdict, after, d, x_cpu, forms_cpu, hip_grads):
    """Parameters after the first clip + Adam step.  Adam's first update is lr * x / (|x| + eps) with x = coef * g +
    wd * p0: an element whose |x| ~ eps = 1e-8 -- a tiny gradient, or a clipped gradient that happens to cancel the
    weight-decay term (0.01 % of the elements here) -- moves by up to lr when x changes in its 7th digit, e.g. because
    the total norm behind the clip coefficient is accumulated in double (HIP) instead of fp32 (torch: 6e-5 apart on
    this model), or because one pooling near-tie was resolved the other way.  helpers.adam_first_step_allowance bounds
    exactly that and nothing else, so:
    (a) the optimizer arithmetic: the oracle's clip + Adam on the HIP gradients vs the HIP parameters, EVERY element,
        3e-6 + the allowance for a 1e-4 uncertainty of the clip coefficient; 99.9 % of the elements within the plain 3e-6;
    (b) fixture G6 (the REAL reference's step), sampled elements: 3e-6 + the allowance for the difference between the
        HIP gradient and the fp32 oracle's gradient; 95 % of the sample within the plain 3e-6."""
    sd0 = torch_state_dict(name)
    ref = O.train_step({k: v.clone() for k, v in sd0.items()}, cfg, x_cpu, forms_cpu, {})
    coef_ref = min(1.0, 5.0 / (ref["total_norm"] + 1e-6))
    sd_a = {k: v.clone() for k, v in sd0.items()}
    clipped = {n: g.clone() for n, g in hip_grads.items()}
    coef_hip = min(1.0, 5.0 / (float(O.clip_grad_norm(clipped, 5.0)) + 1e-6))
    with torch.no_grad():
        O.adam_step(sd_a, clipped, {}, 1e-3, 1e-4)
    worst, n_plain, n_all = 0.0, 0, 0
    for n in names:
        diff = (sdict[n].detach().cpu() - sd_a[n]).abs()
        allowed = adam_first_step_allowance(hip_grads[n], hip_grads[n], sd0[n], coef_hip, coef_hip, 1e-4)
        worst = max(worst, float((diff - allowed).max()))
        n_plain += int((diff <= 3e-6).sum())
        n_all += diff.numel()
    record(f"{name} clip + Adam kernel vs the oracle's Adam on the same gradients [fraction within 3e-6]", n_plain / n_all)
    assert worst <= 0.0, worst
    assert n_plain >= 0.999 * n_all, (n_plain, n_all)

    def smp(t):
        return torch.from_numpy(sample(t, 8)[:8] if t.numel() >= 8 else np.resize(t.detach().cpu().numpy().ravel(), 8))
    allowed = torch.stack([adam_first_step_allowance(smp(ref["grads"][n]), smp(hip_grads[n]), smp(sd0[n]), coef_ref, coef_hip, 1e-4)
                           for n in names])
    # ... capped by a bound that does NOT come from the HIP gradients (a wrong HIP gradient would widen its own
    # allowance): the HIP gradient may sit no further from the fp32 oracle's than 4x the distance between
    # the fp32 and the float64 evaluation of the oracle itself (+ 1e-6 of the gradient's maximum)
    _, g64 = O.loss_and_grads({k: v.clone() for k, v in sd0.items()}, cfg, x_cpu, forms_cpu, PAD, torch.float64)
    cap = []
    for n in names:
        g32, g64n = smp(ref["grads"][n]), smp(g64[n].float())
        slack = 4.0 * (g32 - g64n).abs() + 1e-6 * float(ref["grads"][n].abs().max())
        cap.append(torch.maximum(adam_first_step_allowance(g32, g32 + slack, smp(sd0[n]), coef_ref, coef_ref, 1e-4),
                                 adam_first_step_allowance(g32, g32 - slack, smp(sd0[n]), coef_ref, coef_ref, 1e-4)))
    allowed = torch.minimum(allowed, torch.stack(cap))
    diff = torch.from_numpy(np.abs(after - d["g6_param_sample_after"]))
    record(f"{name} parameters after one step vs fixture G6 [abs, worst of the sample]", float(diff.max()))
    assert float((diff - allowed).max()) <= 0.0, float((diff - allowed).max())
    assert int((diff <= 3e-6).sum()) >= 0.95 * diff.numel(), int((diff <= 3e-6).sum())


@pytest.mark.parametrize("name", ALL)
def test_train_step_vs_reference_golden(name):
    """trainer.py:303-343 for one batch against fixture G6 produced by the REAL reference:
    loss, per-parameter gradient norms, total norm before clipping, parameters after one Adam step."""
    from img2latex_amd.training import TrainStep
    d, cfg, m = build(name)
    big = name in BIG
    T = 24 if big else 12
    forms = torch.from_numpy(synth.make_formulas(4, T, cfg["vocab_size"], seed=777, min_len=5)).to(DEV)
    x = images(cfg, device=DEV)
    names = json.loads(str(d["g6_param_names"]))
    ts = TrainStep(m, lr=1e-3, weight_decay=1e-4, clip_grad_norm=5.0, pad_token_id=PAD, label_smoothing=0.1)
    assert [n for n, _ in m.named_parameters()] == names
    m.train()
    logits = ts.forward_backward(x, forms)
    if "g6_logits" in d:
        abs_close(logits.cpu().numpy(), d["g6_logits"], 1e-4, f"{name} TrainStep logits")
    else:
        abs_close(sample(logits), d["g6_logits_sample"], 1e-4, f"{name} TrainStep logits")
    count = float(ts.flat_grads[ts.n + 1])
    loss = float(ts.flat_grads[ts.n]) / count
    assert count == float((forms[:, 1:] != PAD).sum())
    assert abs(loss - float(d["g6_loss"])) <= 1e-5 * max(1.0, abs(float(d["g6_loss"])))
    # same arithmetic as the fixture (torch CPU fp32 .norm(): its rounding drifts by 6e-4 on the 21M-element FC weight)
    gn = np.array([float((ts.grad_views[n].cpu() / count).norm()) for n in names])
    np.testing.assert_allclose(gn, d["g6_grad_norms"], rtol=3e-4, atol=1e-7)
    ts.apply()
    assert abs(float(ts.stats[0]) - float(d["g6_total_norm"])) <= 3e-4 * float(d["g6_total_norm"])
    sdict = dict(m.named_parameters())
    after = np.stack([sample(sdict[k], 8)[:8] if sdict[k].numel() >= 8 else np.resize(sdict[k].detach().cpu().numpy().ravel(), 8)
                      for k in names])
    _params_after_vs_fixture(name, cfg, names, sdict, after, d, x.cpu(), forms.cpu(),
                             {n: (ts.grad_views[n] / count).cpu() for n in names})
    # the parameters are views into the flat buffer and the state_dict keys are untouched
    assert list(m.state_dict().keys()) == names


def test_train_step_loss_decreases_and_dropout_runs():
    from img2latex_amd.training import TrainStep
    cfg = synth.model_config(vocab_size=50, embedding_dim=32, hidden_dim=64, lstm_layers=2, attention=True, channels=1,
                             img_height=16, img_width=32, conv_filters=(4, 8, 16), dropout=0.3)
    torch.manual_seed(0)
    m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg)).to(DEV)
    ts = TrainStep(m, lr=3e-3, seed=11)
    x = images(cfg, batch=8, device=DEV)
    forms = torch.from_numpy(synth.make_formulas(8, 14, cfg["vocab_size"], seed=5, min_len=6)).to(DEV)
    losses = [float(ts.step(x, forms)["loss"]) for _ in range(30)]
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0], losses[::5]


def test_side_stream_weight_gradients_are_the_same_numbers():
    """Side lanes (i2l_lanes, owned by TrainStep) move the weight-gradient kernels off the main stream: same kernels, same sums, so every
    gradient is bit-identical with and without it (primary dims, 64 x 21) -- except the embedding rows, whose scatter adds
    with atomics in an order that varies from launch to launch on ONE stream as well (compared to 1e-6 of the largest)."""
    from img2latex_amd.training import TrainStep
    runs = []
    for side in (True, False):
        _, cfg, m = build("primary")
        m.train()
        ts = TrainStep(m, seed=3, side_wgrad=side)
        x = images(cfg, batch=64, seed=61, device=DEV)
        forms = torch.from_numpy(synth.make_formulas(64, 22, cfg["vocab_size"], seed=62, min_len=5)).to(DEV)
        ts.forward_backward(x, forms)
        assert not ts._side_keep                               # joined and released
        torch.cuda.synchronize()
        runs.append((ts, ts.flat_grads.clone()))
    ts, a = runs[0]
    b = runs[1][1]
    sizes = {n: p.numel() for n, p in ts.model.named_parameters()}
    for name, off in ts.offsets.items():
        ga, gb = a[off:off + sizes[name]], b[off:off + sizes[name]]
        if name == "decoder.embedding.weight":
            assert float((ga - gb).abs().max()) <= 1e-6 * float(gb.abs().max())
        else:
            assert torch.equal(ga, gb), name
    assert torch.equal(a[ts.n:ts.n + 2], b[ts.n:ts.n + 2])      # loss sum, token count


def test_side_wgrad_flag_through_the_c_abi():
    """i2l_lanes_create / _join / _destroy called the way a C host would: two caller-made streams, conv block backward (with
    data gradient: lane 1) and linear backward (lane 0) with the lanes object, one join, then every output equals the
    one-stream call (lanes == NULL) bit for bit.  Argument errors come back as codes."""
    import ctypes
    torch.manual_seed(9)
    L = _lib.lib()
    side_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    arr = (ctypes.c_void_p * 2)(*[s.cuda_stream for s in side_streams])
    handle = ctypes.c_void_p()
    assert L.i2l_lanes_create(arr, 2, ctypes.byref(handle)) == 0 and handle.value
    assert L.i2l_lanes_create(arr, 0, ctypes.byref(handle)) == -1 and L.i2l_lanes_create(arr, 5, ctypes.byref(handle)) == -1
    assert L.i2l_lanes_create((ctypes.c_void_p * 1)(None), 1, ctypes.byref(ctypes.c_void_p())) == -1     # the default stream
    assert L.i2l_lanes_join(None, _lib.stream_ptr()) == -1 and L.i2l_lanes_destroy(None) == 0
    B, Cin, H, W, Cout = 8, 32, 16, 80, 64
    x = torch.randn(B, Cin, H, W, device=DEV)
    w = torch.randn(Cout, Cin, 3, 3, device=DEV) / 17.0
    b = torch.randn(Cout, device=DEV)
    y = torch.empty(B, Cout, H // 2, W // 2, device=DEV)
    am = torch.empty(y.shape, dtype=torch.uint8, device=DEV)
    nb = L.i2l_conv_workspace_bytes(Cin, Cout)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    assert L.i2l_conv3x3_relu_pool2_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), am.data_ptr(), B, Cin, H, W,
                                        Cout, ws.data_ptr(), nb, 0, _lib.stream_ptr()) == 0
    dy = torch.randn_like(y)
    M, K, N = 64, 2048, 256
    fx, fw = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / 45.0
    fy = torch.relu(fx @ fw.t())
    fdy = torch.randn(M, N, device=DEV)
    outs = []
    for fl in (None, handle.value):
        dx, dw, db = torch.empty_like(x), torch.full_like(w, float("nan")), torch.full_like(b, float("nan"))
        nb2 = L.i2l_conv_bwd_workspace_bytes(B, Cin, H, W, Cout)
        ws2 = torch.empty(nb2, dtype=torch.uint8, device=DEV)
        assert L.i2l_conv3x3_relu_pool2_bwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), am.data_ptr(), dy.data_ptr(),
                                            dx.data_ptr(), dw.data_ptr(), db.data_ptr(), B, Cin, H, W, Cout,
                                            ws2.data_ptr(), nb2, 0, fl, _lib.stream_ptr()) == 0
        fdx, fdw, fdb = torch.empty_like(fx), torch.full_like(fw, float("nan")), torch.full((N,), float("nan"), device=DEV)
        nb3 = L.i2l_linear_bwd_workspace_bytes(M, K, N)
        ws3 = torch.empty(nb3, dtype=torch.uint8, device=DEV)
        assert L.i2l_linear_bias_act_bwd(fx.data_ptr(), fw.data_ptr(), fy.data_ptr(), fdy.data_ptr(), fdx.data_ptr(),
                                         fdw.data_ptr(), fdb.data_ptr(), M, K, N, 1, ws3.data_ptr(), nb3, 0, fl,
                                         _lib.stream_ptr()) == 0
        if fl:
            assert L.i2l_lanes_join(fl, _lib.stream_ptr()) == 0
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (dx, dw, db, fdx, fdw, fdb)])
    assert L.i2l_lanes_destroy(handle.value) == 0
    for a, c in zip(outs[0], outs[1]):
        assert not torch.isnan(c).any()
        assert torch.equal(a, c)


def test_data_parallel_shards_equal_full_batch():
    """SURVEY 8e parity for the DP step, emulated in one process: two ranks' flat buffers are summed
    (what the RCCL all-reduce does), then every rank applies clip + Adam with the GLOBAL count; the
    parameters must equal the single-process full-batch step."""
    from img2latex_amd.training import TrainStep, shard_batch
    name = "tiny_l2_attn"
    _, cfg, m_full = build(name)
    _, _, m_a = build(name)
    _, _, m_b = build(name)
    B, T = 9, 13
    x = images(cfg, batch=B, seed=31, device=DEV)
    forms = torch.from_numpy(synth.make_formulas(B, T, cfg["vocab_size"], seed=32, min_len=4)).to(DEV)
    full, ra, rb = TrainStep(m_full), TrainStep(m_a), TrainStep(m_b)
    full.step(x, forms)
    (lo0, hi0), (lo1, hi1) = shard_batch(B, 0, 2), shard_batch(B, 1, 2)
    for m_ in (m_a, m_b):
        m_.train()
    ra.forward_backward(x[lo0:hi0].contiguous(), forms[lo0:hi0].contiguous())
    rb.forward_backward(x[lo1:hi1].contiguous(), forms[lo1:hi1].contiguous())
    total = ra.flat_grads + rb.flat_grads                      # == all_reduce(SUM)
    ra.flat_grads.copy_(total)
    rb.flat_grads.copy_(total)
    ra.apply()
    rb.apply()
    assert torch.equal(ra.flat_params, rb.flat_params)         # replicas stay bit-identical
    assert float(ra.flat_grads[ra.n + 1]) == float(full.flat_grads[full.n + 1])
    err = float((ra.flat_params - full.flat_params).abs().max())
    assert err <= 1e-5, err                                    # within 1e-5 after one step
    assert abs(float(ra.stats[0]) - float(full.stats[0])) <= 1e-4 * float(full.stats[0])


@pytest.mark.parametrize("amp", [False, True])
@pytest.mark.parametrize("name", ["tiny_l2_attn", "primary"])
def test_reference_trainer_loop_over_the_dropin_model(name, amp):
    """INTEGRATION route A: the reference's own optimisation step (Trainer.train_epoch, trainer.py:303-343), written
    here as its literal torch calls -- nn.CrossEntropyLoss(label_smoothing), loss.backward(), clip_grad_norm_,
    torch.optim.Adam.step -- over the drop-in Seq2SeqModel, against fixture G6 from the REAL reference: loss,
    per-parameter gradient norms, total norm, parameters after the step.  amp=True is the branch the reference takes
    on a ROCm device (device.type == "cuda" => torch.autocast + GradScaler, trainer.py:99-108,312-331): the HIP
    kernels are not autocast-eligible ops, so they keep computing in fp32 (>= the reference's precision) and the
    loss-scaled backward must land on the same parameters."""
    d, cfg, m = build(name)
    big = name in BIG
    T = 24 if big else 12
    formulas = torch.from_numpy(synth.make_formulas(4, T, cfg["vocab_size"], seed=777, min_len=5)).to(DEV)    # int64
    images_ = images(cfg, device=DEV)
    names = json.loads(str(d["g6_param_names"]))
    m.train()
    optimizer = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)                      # trainer.py:91-93
    criterion = torch.nn.CrossEntropyLoss(ignore_index=PAD, reduction="mean", label_smoothing=0.1)    # :111-115
    scaler = torch.amp.GradScaler() if amp else None                                              # :99-105
    optimizer.zero_grad(set_to_none=True)
    targets = formulas[:, 1:]                                                                     # :306
    if amp:
        with torch.autocast(device_type="cuda"):                                                  # :312-316
            outputs = m(images_, formulas)
            logits = outputs.transpose(1, 2)
            loss = criterion(logits, targets)
        assert outputs.dtype == torch.float32
        scaler.scale(loss).backward()                                                             # :320
        scaler.unscale_(optimizer)
    else:
        outputs = m(images_, formulas)                                                            # :334-337
        logits = outputs.transpose(1, 2)
        loss = criterion(logits, targets)
        loss.backward()
    sdict = dict(m.named_parameters())
    gn = np.array([float(sdict[n].grad.cpu().norm()) for n in names])
    raw_grads = {n: sdict[n].grad.detach().cpu().clone() for n in names}
    total = torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)                                   # :338-341 / :323
    if amp:
        scaler.step(optimizer)
        scaler.update()
    else:
        optimizer.step()
    assert abs(float(loss) - float(d["g6_loss"])) <= 1e-5 * max(1.0, abs(float(d["g6_loss"])))
    np.testing.assert_allclose(gn, d["g6_grad_norms"], rtol=3e-4, atol=1e-7)
    assert abs(float(total) - float(d["g6_total_norm"])) <= 3e-4 * float(d["g6_total_norm"])
    after = np.stack([sample(sdict[k], 8)[:8] if sdict[k].numel() >= 8 else np.resize(sdict[k].detach().cpu().numpy().ravel(), 8)
                      for k in names])
    _params_after_vs_fixture(name, cfg, names, sdict, after, d, images_.cpu(), formulas.cpu(),
                             raw_grads)
    optimizer.zero_grad(set_to_none=True)
    # the step changed the parameters through torch: the decoder's cached weight images must notice (p._version)
    m.eval()
    with torch.no_grad():
        a = m(images_, formulas)
        m.decoder._ws_by_stream.clear()
        b = m(images_, formulas)
    assert torch.equal(a, b)


def test_gradient_accumulation_vs_oracle():
    """TrainStep.micro_step = the reference's accumulation branch (trainer.py:345-383; the shipped config has
    accumulation_steps: 4): every micro-batch adds the gradient of ITS mean loss / k, the update comes every k-th batch or
    at the last one of the epoch.  Three micro-batches with k = 2 -- one full group, one partial -- against the oracle:
    per-micro-batch gradients by autograd, summed / k, clipped by the global norm, Adam; parameters within 1e-5 after
    each of the two updates, and nothing moves between updates."""
    from img2latex_amd.training import TrainStep
    cfg = synth.model_config(vocab_size=50, embedding_dim=32, hidden_dim=64, lstm_layers=1, attention=False, channels=1,
                             img_height=16, img_width=32, conv_filters=(4, 8, 16), dropout=0.0)
    np_sd = synth.make_state_dict(cfg, seed=21)
    m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in np_sd.items()})
    m = m.to(DEV)
    ts = TrainStep(m, lr=1e-3, seed=5)
    sd = O.to_torch_sd(np_sd)
    state = {}
    k = 2
    micro = [(torch.from_numpy(synth.make_images(5 + j, cfg, seed=80 + j)),
              torch.from_numpy(synth.make_formulas(5 + j, 12, cfg["vocab_size"], seed=90 + j, min_len=4))) for j in range(3)]
    acc = None
    for j, (x, f) in enumerate(micro):
        before = {n: p.detach().clone() for n, p in m.named_parameters()}
        update = (j + 1) % k == 0 or j == len(micro) - 1
        out = ts.micro_step(x.to(DEV), f.to(DEV), k, update)
        loss, g = O.loss_and_grads(sd, cfg, x, f)
        assert abs(float(out["loss"]) - loss) <= 1e-5 * max(1.0, abs(loss))
        acc = {n: v / k for n, v in g.items()} if acc is None else {n: acc[n] + g[n] / k for n in g}
        if not update:
            for n, p in m.named_parameters():
                assert torch.equal(p.detach(), before[n]), n           # no update inside a group
            continue
        total = O.clip_grad_norm(acc, 5.0)
        with torch.no_grad():
            O.adam_step(sd, acc, state)
        acc = None
        assert abs(float(out["total_norm"]) - float(total)) <= 1e-4 * max(1.0, float(total))
        worst = max(float((p.detach().cpu() - sd[n]).abs().max()) for n, p in m.named_parameters())
        record(f"gradient accumulation k={k}, update {ts.step_count}: parameters vs oracle [abs]", worst)
        assert worst <= 1e-5, worst
    assert ts.step_count == 2 and ts._micro == 0
