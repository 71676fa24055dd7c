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


@pytest.mark.parametrize("name,freeze,B,H,W", [("resnet50", True, 4, 64, 320), ("resnet18", False, 3, 32, 96),
                                               ("resnet50", False, 2, 32, 64)])
def test_resnet_training_forward_backward_vs_oracle(name, freeze, B, H, W):
    enc, sd = _encoder(name, H, W, freeze, seed=11)
    trainable = ["encoder." + n for n, p in enc.named_parameters() if p.requires_grad]
    if freeze:        # encoder.py:201-210: layer4 + the Linear
        assert all(n.startswith(("encoder.resnet.7.", "encoder.embedding_layer.")) for n in trainable) and len(trainable) > 10
    x = torch.from_numpy(synth.uniform(5, "images", (B, 3, H, W), -1.0, 1.0))
    dout = torch.from_numpy(synth.uniform(6, "dout", (B, 64), -1.0, 1.0))
    enc.train()
    out = enc(x.to(DEV))
    assert out.requires_grad
    (out * dout.to(DEV)).sum().backward()
    want_out, want_g, want_stats = RO.resnet_encoder_train_step(sd, name, x, dout, trainable, emulate_bf16=True)
    f32_out, f32_g, _ = RO.resnet_encoder_train_step(sd, name, x, dout, trainable, emulate_bf16=False)
    e_out = float((out.detach().cpu() - want_out).abs().max()) / max(1.0, float(want_out.abs().max()))
    record(f"{name} train fwd B={B} {H}x{W} output vs bf16-emulating oracle [rel to max(1,|ref|)]", e_out)
    record(f"{name} train fwd B={B} {H}x{W} output vs fp32 oracle [rel to max(1,|ref|)]",
           float((out.detach().cpu() - f32_out).abs().max()) / max(1.0, float(f32_out.abs().max())))
    assert e_out <= 3e-2, e_out
    # running statistics of EVERY BatchNorm (frozen ones too), num_batches_tracked
    got_sd = enc.state_dict()
    worst_stat = 0.0
    for k, v in want_stats.items():
        g = got_sd[k[len("encoder."):]].cpu()
        worst_stat = max(worst_stat, float((g - v).abs().max()) / max(1e-3, float(v.abs().max())))
    record(f"{name} train fwd running statistics vs oracle [rel to max]", worst_stat)
    assert worst_stat <= 2e-2, worst_stat
    assert all(int(v) == 1 for k, v in got_sd.items() if k.endswith("num_batches_tracked"))
    # gradients: exactly the trainable set has them
    worst, worst32, low_cos = 0.0, 0.0, 1.0
    for n, p in enc.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, n
            continue
        g, w_, w32 = p.grad.cpu(), want_g["encoder." + n], f32_g["encoder." + n]
        scale = float(w_.abs().max()) + 1e-12
        err = float((g - w_).abs().max()) / scale
        worst, worst32 = max(worst, err), max(worst32, float((g - w32).abs().max()) / (float(w32.abs().max()) + 1e-12))
        low_cos = min(low_cos, _cos(g, w_))
        assert torch.isfinite(g).all(), n
    record(f"{name} freeze={freeze} gradients vs bf16-emulating oracle [worst rel to max]", worst)
    record(f"{name} freeze={freeze} gradients vs fp32 oracle [worst rel to max]", worst32)
    record(f"{name} freeze={freeze} gradients: lowest cosine vs bf16-emulating oracle", 1.0 - low_cos)
    assert low_cos >= 0.995, low_cos
    assert worst <= 8e-2, worst
    # eval mode afterwards uses the UPDATED running statistics (folded-weight cache keyed on their versions)
    enc.eval()
    with torch.no_grad():
        ev = enc(x.to(DEV))
    sd2 = dict(sd)
    sd2.update(want_stats)
    want_ev = RO.resnet_encoder(sd2, name, x)
    assert float((ev.cpu() - want_ev).abs().max()) <= 4e-2 * max(1.0, float(want_ev.abs().max()))
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
