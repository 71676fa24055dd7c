"""ResNetEncoder in TRAINING mode (row a3 of SURVEY 8: encoder.py:185-249 under model.train()): BatchNorm batch
statistics + running-statistic updates, `freeze_backbone` semantics (:201-210: only layer4 and the Linear get
gradients; the shipped config trains everything, configs/config.yaml:43), the trunk's backward on the HIP kernels of
csrc/resnet_train.hip, through torch.autograd (the reference Trainer's loss.backward()) and through TrainStep.

Parity stays UNPINNED against the reference itself (torchvision absent, remote weights: SURVEY 8c).  The yardstick is
oracle/resnet_oracle.py's autograd restatement, evaluated twice: with bf16 rounding emulated at the points where the
HIP path stores bf16 (isolates kernel defects from the precision of the bf16 data path) and in plain fp32 (what the
reference's modules compute; the distance is dominated by bf16 activations and recorded)."""
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
    this repo's kernels), fed the operands the HIP unit saw (bf16-rounded input and filter).  With dy: also the
    gradients of sum(y * dy) with respect to input, filter, gamma, beta and the residual."""
    conv, bn = u["conv"], u["bn"]
    x = u["x"].detach()
    x = _bf(x) if u["nchw"] else x.float().permute(0, 3, 1, 2)
    x = x.contiguous().requires_grad_(dy is not None)
    w = _bf(conv.weight.detach()).requires_grad_(dy is not None)
    gamma, beta = (t.detach().clone().requires_grad_(dy is not None) for t in (bn.weight, bn.bias))
    res = None
    if u["residual"] is not None:
        res = u["residual"].float().permute(0, 3, 1, 2).contiguous().requires_grad_(dy is not None)
    z = torch.nn.functional.conv2d(x, w, stride=conv.stride, padding=conv.padding)
    yn = torch.nn.functional.batch_norm(z, None, None, gamma, beta, training=True, eps=bn.eps)
    y = yn if res is None else yn + res
    if u["relu"] and dy is not None:
        # the ReLU gate is a discrete decision: an activation within bf16 rounding of zero may fall either way, and a
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


@pytest.mark.parametrize("name,freeze,B,H,W", [("resnet50", True, 4, 64, 320), ("resnet18", False, 3, 32, 96),
                                               ("resnet50", False, 3, 64, 128)])
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
    # ---- (1) every unit of the forward against an independent fp32 computation on the same operands
    worst = dict(z=0.0, y=0.0, stat=0.0, run=0.0)
    for u in tape["units"]:
        ref = _unit_reference(u)
        zf, yf = u["z"].float(), u["y"].float()
        # z and y are stored in bf16: one ulp (2^-8 of the value) plus fp32 summation noise
        assert float(((zf - ref["z"]).abs() - 2.0 ** -7 * ref["z"].abs()).max()) <= 2e-3 * max(1.0, float(ref["z"].abs().max()))
        worst["z"] = max(worst["z"], _rel(zf, ref["z"]))
        worst["stat"] = max(worst["stat"], _rel(u["mean"], zf.mean(dim=(0, 1, 2))),
                            _rel(1.0 / u["invstd"] ** 2 - u["bn"].eps, zf.var(dim=(0, 1, 2), unbiased=False)))
        # y from the HIP z / statistics themselves (the statistics of a bf16-rounded z differ slightly from the fp32 z's)
        yn = (zf - u["mean"]) * u["invstd"] * u["bn"].weight.detach() + u["bn"].bias.detach()
        mag = yn.abs() + (u["residual"].float().abs() if u["residual"] is not None else 0.0)   # one bf16 ulp of either term
        yn = _bf(yn) + (u["residual"].float() if u["residual"] is not None else 0.0)
        yn = torch.relu(yn) if u["relu"] else yn
        # two bf16 roundings (the normalised value, then the sum), the first of which may fall on the neighbouring bf16
        # when the fp32 expression is evaluated in another order: <= 1.5 ulp(norm) + 0.5 ulp(y), ulp(v) <= 2^-7 |v|
        assert float(((yf - yn).abs() - 2.0 ** -6 * mag).max()) <= 1e-5 * max(1.0, float(yn.abs().max()))
        worst["y"] = max(worst["y"], _rel(yf, ref["y"]))
        key = next(n for n, mod in enc.named_modules() if mod is u["bn"])
        M = zf.numel() // zf.shape[-1]
        rm = 0.9 * old_stats[key + ".running_mean"] + 0.1 * zf.mean(dim=(0, 1, 2))
        rv = 0.9 * old_stats[key + ".running_var"] + 0.1 * zf.var(dim=(0, 1, 2), unbiased=M > 1)
        worst["run"] = max(worst["run"], _rel(u["bn"].running_mean, rm), _rel(u["bn"].running_var, rv))
        assert int(u["bn"].num_batches_tracked) == 1
    assert worst["stat"] <= 1e-4 and worst["run"] <= 1e-4, worst
    assert worst["z"] <= 1.5e-2 and worst["y"] <= 5e-2, worst
    for k, v in worst.items():
        record(f"{name} B={B} {H}x{W} train forward, worst unit: {k} vs independent fp32 [rel to max]", v)
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
        record(f"{name} freeze={freeze} train backward, worst unit: {k} vs fp32 autograd of the unit [rel to max]", v)
    # the HIP backward reads the bf16-rounded z and y (ReLU mask, xhat): a few 1e-3 of the maximum per unit
    assert max(wb.values()) <= 3e-2, wb
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
    # ---- (3) end to end against the oracle (trainer's view): bf16 differences compound through up to 53 BatchNorms
    #          with batch statistics of as few as 6 samples per channel, so only direction and scale are asserted
    want_out, want_g, want_stats = RO.resnet_encoder_train_step(sd, name, x, dout, trainable, emulate_bf16=True)
    e_out = float((out.detach().cpu() - want_out).abs().max()) / max(1.0, float(want_out.abs().max()))
    record(f"{name} train fwd B={B} {H}x{W} output vs bf16-emulating oracle, end to end [rel to max(1,|ref|)]", e_out)
    cosines = []
    for n, p in enc.named_parameters():
        if p.requires_grad:
            assert torch.isfinite(grads[n]).all(), n
            cosines.append(_cos(grads[n].cpu(), want_g["encoder." + n]))
    cosines = np.sort(np.array(cosines))
    record(f"{name} freeze={freeze} gradients end to end: 1 - lowest cosine vs bf16-emulating oracle", 1.0 - cosines[0])
    record(f"{name} freeze={freeze} gradients end to end: 1 - median cosine vs bf16-emulating oracle", 1.0 - float(np.median(cosines)))
    # End to end the two bf16 data paths drift apart by ~7 % per unit (profiles/r03/resnet_train_diag.txt: steady growth
    # from 3e-3 at the stem to 0.3 at layer4, no jump at any layer): random weights, BatchNorm re-normalising every
    # layer with batch statistics over as few as 24 positions and the ReLU gates make the net amplify rounding noise, and
    # two roundings of the same z (fp32 sums in another order) are two different noises.  So end to end only sanity is
    # asserted -- finite, same direction for the bulk of the parameters -- and the unit-level checks (1), (2) and the
    # exact routing check (2b) above carry the correctness claim.
    assert np.isfinite(e_out) and float(np.median(cosines)) >= 0.3, (e_out, cosines[:5], float(np.median(cosines)))
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
