"""ResNetEncoder in TRAINING mode (row a3 of SURVEY 8: encoder.py:185-249 under model.train()): BatchNorm batch
statistics + running-statistic updates, `freeze_backbone` semantics (:201-210: only layer4 and the Linear get
gradients; the shipped config trains everything, configs/config.yaml:43), the trunk's backward on the HIP kernels of
csrc/resnet_train.hip, through torch.autograd (the reference Trainer's loss.backward()) and through TrainStep.

Parity stays UNPINNED against the reference itself (torchvision absent, remote weights: SURVEY 8c).  The yardstick is
oracle/resnet_oracle.py's autograd restatement in plain fp32 -- what the reference's modules compute in the Trainer's
fp32 branch.  Since r04 the HIP training path is fp32 grade too (fp32 activations, split-bf16 matrix-core GEMMs), so
the end-to-end comparison is a real assertion: r03's bf16 tape could only be held to "median cosine >= 0.3"."""
import numpy as np
import pytest
import torch

import resnet_oracle as RO
from conftest import record
from img2latex_amd import _lib, synth
from img2latex_amd.model import ResNetEncoder, Seq2SeqModel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _encoder(name, h, w, freeze, seed):
    enc = ResNetEncoder(h, w, 3, model_name=name, embedding_dim=64, freeze_backbone=freeze)
    shapes = [(k, tuple(v.shape)) for k, v in enc.state_dict().items()]
    np_sd = synth.make_resnet_state_dict(shapes, seed=seed)
    enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in np_sd.items()}, strict=True)
    return enc.to(DEV), {"encoder." + k: torch.from_numpy(v.copy()) for k, v in np_sd.items()}


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _unit_reference(u, dy=None):
    """One conv + BatchNorm(train) (+ residual) (+ ReLU) unit in fp32 torch ON THE DEVICE (rocBLAS / MIOpen, none of
    this repo's kernels), fed the operands the HIP unit saw.  With dy: also the gradients of sum(y * dy) with respect
    to input, filter, gamma, beta and the residual."""
    conv, bn = u["conv"], u["bn"]
    x = u["x"].detach()
    x = x if u["nchw"] else x.permute(0, 3, 1, 2)
    x = x.contiguous().requires_grad_(dy is not None)
    w = conv.weight.detach().clone().requires_grad_(dy is not None)
    gamma, beta = (t.detach().clone().requires_grad_(dy is not None) for t in (bn.weight, bn.bias))
    res = None
    if u["residual"] is not None:
        res = u["residual"].float().permute(0, 3, 1, 2).contiguous().requires_grad_(dy is not None)
    z = torch.nn.functional.conv2d(x, w, stride=conv.stride, padding=conv.padding)
    yn = torch.nn.functional.batch_norm(z, None, None, gamma, beta, training=True, eps=bn.eps)
    y = yn if res is None else yn + res
    if u["relu"] and dy is not None:
        # the ReLU gate is a discrete decision: an activation within fp32 rounding of zero may fall either way, and a
        # flipped gate moves that element's gradient by 100 % -- the gradients are judged UNDER THE HIP FORWARD'S GATES
        # (like the conv blocks' pooling decisions in test_hip_training.py); the forward check above bounds the values
        y = y * (u["y"].float() > 0).permute(0, 3, 1, 2)
    elif u["relu"]:
        y = torch.relu(y)
    out = dict(z=z.detach().permute(0, 2, 3, 1), y=y.detach().permute(0, 2, 3, 1),
               mean=z.detach().mean(dim=(0, 2, 3)), var=z.detach().var(dim=(0, 2, 3), unbiased=False))
    if dy is not None:
        (y * dy.permute(0, 3, 1, 2)).sum().backward()
        out.update(dx=x.grad if u["nchw"] else x.grad.permute(0, 2, 3, 1), dw=w.grad, dgamma=gamma.grad, dbeta=beta.grad,
                   dres=None if res is None else res.grad.permute(0, 2, 3, 1))
    return out


def _rel(a, b):
    return float((a.float() - b.float()).abs().max()) / (float(b.float().abs().max()) + 1e-20)


@pytest.mark.parametrize("name,freeze,B,H,W", [("resnet50", True, 4, 64, 320), ("resnet50", False, 4, 64, 320),
                                               ("resnet50", True, 32, 64, 320), ("resnet50", False, 32, 64, 320),
                                               ("resnet18", False, 3, 32, 96)])
def test_resnet_training_forward_backward_vs_oracle(name, freeze, B, H, W):
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    enc, sd = _encoder(name, H, W, freeze, seed=11)
    trainable = ["encoder." + n for n, p in enc.named_parameters() if p.requires_grad]
    if freeze:        # encoder.py:201-210: layer4 + the Linear
        assert all(n.startswith(("encoder.resnet.7.", "encoder.embedding_layer.")) for n in trainable) and len(trainable) > 10
    x = torch.from_numpy(synth.uniform(5, "images", (B, 3, H, W), -1.0, 1.0))
    dout = torch.from_numpy(synth.uniform(6, "dout", (B, 64), -1.0, 1.0))
    old_stats = {k: v.clone() for k, v in enc.state_dict().items() if "running" in k}
    enc.train()
    from img2latex_amd.model._train_fn import encoder_train_backward, encoder_train_forward
    out, tape = encoder_train_forward(enc, x.to(DEV))
    grads = {n: torch.zeros_like(p) for n, p in enc.named_parameters() if p.requires_grad}
    enc.trace_bwd = []
    encoder_train_backward(enc, tape, dout.to(DEV), grads)
    bwd, enc.trace_bwd = enc.trace_bwd, None
    tag = f"{name} freeze={freeze} B={B} {H}x{W}"
    # ---- (1) every unit of the forward against an independent fp32 computation on the same operands
    worst = dict(z=0.0, y=0.0, stat=0.0, run=0.0)
    for u in tape["units"]:
        assert u["z"].dtype == torch.float32 and u["y"].dtype == torch.float32       # the tape is fp32 (r04)
        ref = _unit_reference(u)
        zf, yf = u["z"], u["y"]
        worst["z"] = max(worst["z"], _rel(zf, ref["z"]))
        # statistics as the normalisation sees them, per channel: the mean's error in units of the channel's spread, the
        # variance's error relative to that variance (sharper than "relative to the largest channel", and well-posed for the
        # stem, whose means are ~1e-3 of its spreads)
        m_ref, v_ref = zf.double().mean(dim=(0, 1, 2)), zf.double().var(dim=(0, 1, 2), unbiased=False)
        e_mean = float(((u["mean"].double() - m_ref).abs() / (v_ref + u["bn"].eps).sqrt()).max())
        e_var = float((((1.0 / u["invstd"].double() ** 2 - u["bn"].eps) - v_ref).abs() / (v_ref + u["bn"].eps)).max())
        worst["stat"] = max(worst["stat"], e_mean, e_var)
        # y from the HIP z / statistics themselves: the elementwise formula, evaluated in another order
        yn = (zf - u["mean"]) * u["invstd"] * u["bn"].weight.detach() + u["bn"].bias.detach()
        yn = yn + (u["residual"] if u["residual"] is not None else 0.0)
        yn = torch.relu(yn) if u["relu"] else yn
        assert float((yf - yn).abs().max()) <= 1e-5 * max(1.0, float(yn.abs().max()))
        worst["y"] = max(worst["y"], _rel(yf, ref["y"]))
        key = next(n for n, mod in enc.named_modules() if mod is u["bn"])
        M = zf.numel() // zf.shape[-1]
        rm = 0.9 * old_stats[key + ".running_mean"].double() + 0.1 * zf.double().mean(dim=(0, 1, 2))
        rv = 0.9 * old_stats[key + ".running_var"].double() + 0.1 * zf.double().var(dim=(0, 1, 2), unbiased=M > 1)
        worst["run"] = max(worst["run"], _rel(u["bn"].running_mean, rm), _rel(u["bn"].running_var, rv))
        assert int(u["bn"].num_batches_tracked) == 1
    for k, v in worst.items():
        record(f"{tag} train forward, worst unit: {k} vs independent fp32 [rel to max; stat: per channel, mean in units of std]", v)
    assert worst["stat"] <= 1e-6 and worst["run"] <= 2e-6, worst          # r03 (statistics of a bf16 z): 1e-4 of the maximum
    assert worst["z"] <= 2e-5 and worst["y"] <= 1e-4, worst               # r03 (bf16 tape): 1.5e-2 / 5e-2
    # ---- (2) every unit of the backward: gradients of the SAME upstream gradient through an fp32 autograd of the unit
    first_trainable = min(i for i, u in enumerate(tape["units"])
                          if any(p.requires_grad for p in list(u["conv"].parameters()) + list(u["bn"].parameters())))
    assert len(bwd) == len(tape["units"]) - first_trainable
    wb = dict(dw=0.0, dx=0.0, dgamma=0.0, dbeta=0.0, dres=0.0)
    for rec in bwd:
        u = rec["unit"]
        ref = _unit_reference(u, rec["dy"])
        if u["conv"].weight.requires_grad:
            wb["dw"] = max(wb["dw"], _rel(grads[rec["conv_name"] + ".weight"], ref["dw"]))
            wb["dgamma"] = max(wb["dgamma"], _rel(grads[rec["bn_name"] + ".weight"], ref["dgamma"]))
            wb["dbeta"] = max(wb["dbeta"], _rel(grads[rec["bn_name"] + ".bias"], ref["dbeta"]))
        if rec["dx"] is not None:
            wb["dx"] = max(wb["dx"], _rel(rec["dx"], ref["dx"]))
        if rec["dres"] is not None and ref["dres"] is not None:
            wb["dres"] = max(wb["dres"], _rel(rec["dres"], ref["dres"]))
    for k, v in wb.items():
        record(f"{tag} train backward, worst unit: {k} vs fp32 autograd of the unit [rel to max]", v)
    assert max(wb.values()) <= 1e-3, wb                                   # r03 (bf16 z / y on the tape): 3e-2
    # ---- (2b) the ROUTING between the units, exactly: every unit's upstream gradient is the sum of what its consumers
    #           produced (main path, residual branch / downsample), the head of the chain is the average-pool backward of
    #           d(features), the stem's is the max-pool backward (checked against torch's on the same input)
    by_unit = {id(r["unit"]): r for r in bwd}
    units, blocks = tape["units"], tape["blocks"]
    Bf, Hf, Wf, Cf = tape["final_shape"]
    dfeat_chain = by_unit[id(units[blocks[-1]["main"][-1]])]["dy"]
    assert torch.equal(dfeat_chain, dfeat_chain[:, :1, :1, :].expand(Bf, Hf, Wf, Cf))       # constant over positions
    for bi in reversed(range(len(blocks))):
        rec = blocks[bi]
        main, down = rec["main"], rec["down"]
        if id(units[main[-1]]) not in by_unit:
            break
        r_last = by_unit[id(units[main[-1]])]
        for a, b in zip(main[:-1], main[1:]):                    # main path: dy of a unit IS the dx of the next one
            if id(units[a]) in by_unit:
                assert by_unit[id(units[a])]["dy"] is by_unit[id(units[b])]["dx"]
        if down is not None and id(units[down]) in by_unit:
            assert by_unit[id(units[down])]["dy"] is r_last["dres"]
        if bi > 0 and id(units[blocks[bi - 1]["main"][-1]]) in by_unit:
            r_first = by_unit[id(units[main[0]])]
            want_dh = r_first["dx"] + (by_unit[id(units[down])]["dx"] if down is not None else r_last["dres"])
            assert torch.equal(by_unit[id(units[blocks[bi - 1]["main"][-1]])]["dy"], want_dh), bi
    if first_trainable == 0:                                     # stem: max-pool backward of the first block's input gradient
        r0 = by_unit[id(units[0])]
        b0 = blocks[0]
        r_first, r_last0 = by_unit[id(units[b0["main"][0]])], by_unit[id(units[b0["main"][-1]])]
        dh0 = r_first["dx"] + (by_unit[id(units[b0["down"]])]["dx"] if b0["down"] is not None else r_last0["dres"])
        pin = tape["pool_in"].float().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        torch.nn.functional.max_pool2d(pin, 3, stride=2, padding=1).backward(dh0.permute(0, 3, 1, 2).contiguous())
        assert torch.equal(r0["dy"], pin.grad.permute(0, 2, 3, 1).contiguous())
    # ---- (3) end to end against the fp32 oracle (the Trainer's view).  The random-weight trunk amplifies a relative
    #          perturbation ~100x over its 53 units (profiles/micro/resnet_precision_cpu.py: the fp32 oracle itself sits
    #          at cosine 0.9991 - 0.9998 from its own float64 evaluation), so the bound is a cosine, for EVERY parameter
    want_out, want_g, want_stats = RO.resnet_encoder_train_step(sd, name, x, dout, trainable, emulate_bf16=False)
    e_out = float((out.detach().cpu() - want_out).abs().max()) / max(1.0, float(want_out.abs().max()))
    record(f"{tag} train fwd output vs fp32 oracle, end to end [rel to max(1,|ref|)]", e_out)
    cosines = {}
    for n, p in enc.named_parameters():
        if p.requires_grad:
            assert torch.isfinite(grads[n]).all(), n
            cosines[n] = _cos(grads[n].cpu(), want_g["encoder." + n])
    cs = np.sort(np.array(list(cosines.values())))
    record(f"{tag} gradients end to end: 1 - lowest cosine vs fp32 oracle", 1.0 - cs[0])
    record(f"{tag} gradients end to end: 1 - median cosine vs fp32 oracle", 1.0 - float(np.median(cs)))
    e_stat = max(_rel(enc.state_dict()[k[len("encoder."):]].cpu(), v) for k, v in want_stats.items())
    record(f"{tag} running statistics after the step vs fp32 oracle [rel to max]", e_stat)
    assert e_out <= 3e-2, e_out
    assert cs[0] >= 0.99, sorted(cosines.items(), key=lambda kv: kv[1])[:5]
    assert e_stat <= 1e-2, e_stat
    # ---- (4) the torch.autograd route (the reference Trainer's loss.backward()) gives the same gradients; frozen
    #          parameters get none
    enc2, _ = _encoder(name, H, W, freeze, seed=11)
    enc2.train()
    o2 = enc2(x.to(DEV))
    assert o2.requires_grad and torch.equal(o2.detach(), out)
    (o2 * dout.to(DEV)).sum().backward()
    for n, p in enc2.named_parameters():
        if p.requires_grad:
            assert torch.equal(p.grad, grads[n]), n
        else:
            assert p.grad is None, n
    # ---- (5) eval mode afterwards uses the UPDATED running statistics (folded-weight cache keyed on their versions)
    enc.eval()
    with torch.no_grad():
        ev = enc(x.to(DEV))
    sd2 = {"encoder." + k: v.detach().cpu() for k, v in enc.state_dict().items()}
    want_ev = RO.resnet_encoder(sd2, name, x)
    assert float((ev.cpu() - want_ev).abs().max()) <= 6e-2 * max(1.0, float(want_ev.abs().max()))
    with pytest.raises(NotImplementedError):
        enc(x.to(DEV))                                   # eval mode + gradients: not the reference's training flow


def test_resnet_lstm_train_step():
    """TrainStep (trainer.py:303-343 on the HIP kernels) over a resnet_lstm model: only layer4 / Linear / decoder are
    optimised (frozen parameters get no update and no weight decay, like torch's Adam skipping .grad None), the loss
    goes down over a few steps, and the torch-autograd route (loss.backward(), as the reference's Trainer) gives the
    same first-step gradients."""
    from img2latex_amd.training import TrainStep
    cfg = synth.model_config(vocab_size=60, embedding_dim=64, hidden_dim=64, dropout=0.0)
    enc_p = dict(img_height=32, img_width=96, channels=3, model_name="resnet18", embedding_dim=64, freeze_backbone=True)

    def build():
        m = Seq2SeqModel("resnet_lstm", 60, enc_p, synth.decoder_params(cfg))
        shapes = [(k, tuple(v.shape)) for k, v in m.encoder.state_dict().items()]
        full = {"encoder." + k: torch.from_numpy(v) for k, v in synth.make_resnet_state_dict(shapes, seed=3).items()}
        full.update({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=4).items() if k.startswith("decoder.")})
        m.load_state_dict(full)
        return m.to(DEV)

    x = torch.from_numpy(synth.uniform(8, "images", (6, 3, 32, 96), -1.0, 1.0)).to(DEV)
    forms = torch.from_numpy(synth.make_formulas(6, 14, 60, seed=9, min_len=5)).to(DEV)
    m = build()
    frozen_before = {n: p.detach().clone() for n, p in m.named_parameters() if not p.requires_grad}
    assert frozen_before
    ts = TrainStep(m, lr=1e-3)
    assert set(ts.offsets) == {n for n, p in m.named_parameters() if p.requires_grad}
    losses = [float(ts.step(x, forms)["loss"]) for _ in range(6)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    for n, p in m.named_parameters():
        if not p.requires_grad:
            assert torch.equal(p.detach(), frozen_before[n]), n
    osd = ts.optimizer_state_dict()
    n_all = len(list(m.parameters()))
    assert osd["param_groups"][0]["params"] == list(range(n_all)) and len(osd["state"]) == len(ts.offsets)
    # the autograd route on a fresh copy: same gradients as TrainStep's first forward_backward
    a, b = build(), build()
    tsb = TrainStep(b)
    b.train()
    tsb.forward_backward(x, forms)
    count = float(tsb.flat_grads[tsb.n + 1])
    a.train()
    crit = torch.nn.CrossEntropyLoss(ignore_index=0, reduction="mean", label_smoothing=0.1)
    loss = crit(a(x, forms).transpose(1, 2), forms[:, 1:])
    loss.backward()
    for n, p in a.named_parameters():
        if p.requires_grad:
            g2 = tsb.grad_views[n] / count
            assert float((p.grad - g2).abs().max()) <= 1e-5 * max(1.0, float(g2.abs().max())), n


def test_resnet50_lstm_30_step_loss_curve_vs_fp32_oracle():
    """30 optimisation steps of an UNFROZEN resnet50_lstm (the shipped configuration trains the trunk,
    configs/config.yaml:43) through TrainStep, every one of them held against the oracle's fp32 step (trainer.py:303-343
    restated: oracle/resnet_oracle.py::resnet_lstm_train_step) FROM THE SAME STATE on the same batch: before its step i
    the oracle takes over the parameters, Adam moments and running statistics TrainStep has at that point, so the
    comparison is 30 different states along a real training trajectory and none of it is the trajectory's own chaos.
    Why not two free-running curves: 30 Adam steps through this random-weight trunk amplify rounding noise until two CORRECT
    evaluations -- the oracle in float32 and in float64 -- are 1.2 % apart in loss and 0.99 of the maximum apart in running
    statistics (profiles/micro/resnet_chaos_cpu.py); against the free-running fp32 oracle the HIP curve measured 1.3 % on
    one box and 2.2 % on another, i.e. the noise floor, not the kernels.  That free-running distance is still recorded.
    Bounds (same state, same batch): loss 1e-4 relative (VERDICT r03 asked 2 % of a free-running curve), total gradient
    norm 2e-3, and the parameter UPDATE of every step points the oracle's way (cosine >= 0.98 over all parameters: Adam's
    normalised step turns a gradient element near zero into +-lr, measured 1 - cos = 8e-3 at worst)."""
    from img2latex_amd.training import TrainStep
    cfg = synth.model_config(vocab_size=60, embedding_dim=64, hidden_dim=64, dropout=0.0)
    enc_p = dict(img_height=64, img_width=320, channels=3, model_name="resnet50", embedding_dim=64, freeze_backbone=False)
    m = Seq2SeqModel("resnet_lstm", 60, enc_p, synth.decoder_params(cfg))
    shapes = [(k, tuple(v.shape)) for k, v in m.encoder.state_dict().items()]
    full = {"encoder." + k: torch.from_numpy(v) for k, v in synth.make_resnet_state_dict(shapes, seed=3).items()}
    full.update({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=4).items() if k.startswith("decoder.")})
    m.load_state_dict(full)
    m = m.to(DEV)
    trainable = [n for n, p in m.named_parameters() if p.requires_grad]
    params = dict(m.named_parameters())
    B, steps = 8, 30
    batches = [(torch.from_numpy(synth.uniform(20 + i, "images", (B, 3, 64, 320), -1.0, 1.0)),
                torch.from_numpy(synth.make_formulas(B, 14, 60, seed=30 + i, min_len=5))) for i in range(4)]
    ts = TrainStep(m, lr=1e-3)
    free_sd, free_state = {k: v.clone() for k, v in full.items()}, {}
    got, worst = [], dict(loss=0.0, norm=0.0, one_minus_cos=0.0, free=0.0)
    for i in range(steps):
        x, f = batches[i % len(batches)]
        # the oracle's state := TrainStep's state before step i
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        state = {"step": ts.step_count}
        for n in trainable:
            o, k = ts.offsets[n], params[n].numel()
            state["m." + n] = ts.exp_avg[o:o + k].view_as(params[n]).detach().cpu().clone()
            state["v." + n] = ts.exp_avg_sq[o:o + k].view_as(params[n]).detach().cpu().clone()
        before = torch.cat([sd[n].reshape(-1) for n in trainable]).double()
        out = ts.step(x.to(DEV), f.to(DEV))
        got.append(float(out["loss"]))
        ref = RO.resnet_lstm_train_step(sd, "resnet50", cfg, x, f, state, set(trainable), lr=1e-3)
        worst["loss"] = max(worst["loss"], abs(got[-1] - ref["loss"]) / abs(ref["loss"]))
        worst["norm"] = max(worst["norm"], abs(float(out["total_norm"]) - ref["total_norm"]) / ref["total_norm"])
        d_hip = torch.cat([params[n].detach().cpu().reshape(-1) for n in trainable]).double() - before
        d_ref = torch.cat([sd[n].reshape(-1) for n in trainable]).double() - before
        worst["one_minus_cos"] = max(worst["one_minus_cos"], 1.0 - _cos(d_hip, d_ref))
        if i == 0:                                                       # running statistics after one step from equal states
            rs = max(_rel(m.state_dict()[k].cpu(), sd[k]) for k in sd if "running" in k)
            assert rs <= 1e-4, rs
        # the free-running oracle curve, for the record
        fr = RO.resnet_lstm_train_step(free_sd, "resnet50", cfg, x, f, free_state, set(trainable), lr=1e-3)["loss"]
        worst["free"] = max(worst["free"], abs(got[-1] - fr) / fr)
    record("resnet50_lstm unfrozen, 30 TrainStep steps vs the fp32 oracle step from the same state: worst loss difference [rel]", worst["loss"])
    record("resnet50_lstm unfrozen, 30 steps: worst total gradient norm difference [rel]", worst["norm"])
    record("resnet50_lstm unfrozen, 30 steps: worst 1 - cosine of the parameter update (all parameters)", worst["one_minus_cos"])
    record("resnet50_lstm unfrozen, 30 steps: worst loss difference to the FREE-RUNNING fp32 oracle [rel; fp32 vs fp64 oracle: 1.2e-2]", worst["free"])
    assert np.isfinite(got).all() and got[-1] < got[0], got
    assert worst["loss"] <= 1e-4 and worst["norm"] <= 2e-3 and worst["one_minus_cos"] <= 2e-2, worst
