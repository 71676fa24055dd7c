"""CPU oracle for the ResNet encoder -- TEST INFRASTRUCTURE, NOT PRODUCT.

PARITY UNPINNED: torchvision is not installed and the reference's ResNetEncoder fetches remote
ImageNet weights (encoder.py:185-194), so no output of the reference itself exists for
this path.  This is a functional fp32 restatement of the published torchvision ResNet v1.5 forward
(conv7x7/2 + BN + ReLU + maxpool3x3/2; BasicBlock / Bottleneck stages with the stride on the 3x3;
adaptive average pool) in the child order the reference slices (encoder.py:198-206), followed by
Flatten + Linear + ReLU (encoder.py:242-247), driven by a state_dict with the reference's key names.
"""
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BLOCKS = {"resnet18": ("basic", [2, 2, 2, 2]), "resnet34": ("basic", [3, 4, 6, 3]),
          "resnet50": ("bottleneck", [3, 4, 6, 3]), "resnet101": ("bottleneck", [3, 4, 23, 3]),
          "resnet152": ("bottleneck", [3, 8, 36, 3])}


def _bn(sd: SD, key: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        training=False, eps=1e-5)


def resnet_trunk(sd: SD, model_name: str, x: torch.Tensor, prefix: str = "encoder.resnet.") -> torch.Tensor:
    kind, counts = BLOCKS[model_name]
    x = F.relu(_bn(sd, prefix + "1", F.conv2d(x, sd[prefix + "0.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    for li, n in enumerate(counts):
        for bi in range(n):
            p = f"{prefix}{4 + li}.{bi}."
            stride = 2 if (li > 0 and bi == 0) else 1
            identity = x
            if kind == "bottleneck":
                o = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"])))
                o = F.relu(_bn(sd, p + "bn2", F.conv2d(o, sd[p + "conv2.weight"], stride=stride, padding=1)))
                o = _bn(sd, p + "bn3", F.conv2d(o, sd[p + "conv3.weight"]))
            else:
                o = F.relu(_bn(sd, p + "bn1", F.conv2d(x, sd[p + "conv1.weight"], stride=stride, padding=1)))
                o = _bn(sd, p + "bn2", F.conv2d(o, sd[p + "conv2.weight"], padding=1))
            if p + "downsample.0.weight" in sd:
                identity = _bn(sd, p + "downsample.1", F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride))
            x = F.relu(o + identity)
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


def resnet_encoder(sd: SD, model_name: str, x: torch.Tensor) -> torch.Tensor:
    feat = resnet_trunk(sd, model_name, x)
    return F.relu(F.linear(feat, sd["encoder.embedding_layer.weight"], sd["encoder.embedding_layer.bias"]))


# ---------------------------------------------------------------------------------------------------------------
# Training mode (encoder.py:185-249 under model.train()): BatchNorm with BATCH statistics + running-statistic
# updates, gradients by torch autograd.  PARITY UNPINNED like the forward.  `emulate_bf16=True` rounds to bf16 at the
# points where the HIP path stores bf16 (conv operands, raw conv output z, the normalised value, the activation) with a
# straight-through gradient, so that a comparison isolates kernel defects from the precision of the bf16 data path;
# `emulate_bf16=False` is the plain fp32 (or fp64) computation of the reference's modules.
# ---------------------------------------------------------------------------------------------------------------
def _rnd(t: torch.Tensor, on: bool) -> torch.Tensor:
    if not on:
        return t
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def _bn_train(sd: SD, key: str, z: torch.Tensor, new_stats: Dict[str, torch.Tensor], emulate: bool) -> torch.Tensor:
    rm, rv = sd[key + ".running_mean"].clone().to(z.dtype), sd[key + ".running_var"].clone().to(z.dtype)
    y = F.batch_norm(z, rm, rv, sd[key + ".weight"], sd[key + ".bias"], training=True, momentum=0.1, eps=1e-5)
    new_stats[key + ".running_mean"], new_stats[key + ".running_var"] = rm, rv
    return _rnd(y, emulate)


def resnet_trunk_train(sd: SD, model_name: str, x: torch.Tensor, new_stats: Dict[str, torch.Tensor],
                       emulate_bf16: bool = False, prefix: str = "encoder.resnet.", taps=None) -> torch.Tensor:
    """`taps`: a dict that receives the activation after every unit, keyed by the conv's state_dict prefix."""
    kind, counts = BLOCKS[model_name]
    e = emulate_bf16
    def tap(key, t):
        if taps is not None:
            taps[key] = t.detach()
        return t
    conv = lambda t, w, **kw: _rnd(F.conv2d(_rnd(t, e), _rnd(w, e), **kw), e)
    x = tap(prefix + "0", F.relu(_bn_train(sd, prefix + "1", conv(x, sd[prefix + "0.weight"], stride=2, padding=3), new_stats, e)))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    for li, n in enumerate(counts):
        for bi in range(n):
            p = f"{prefix}{4 + li}.{bi}."
            stride = 2 if (li > 0 and bi == 0) else 1
            identity = x
            if kind == "bottleneck":
                o = tap(p + "conv1", F.relu(_bn_train(sd, p + "bn1", conv(x, sd[p + "conv1.weight"]), new_stats, e)))
                o = tap(p + "conv2", F.relu(_bn_train(sd, p + "bn2", conv(o, sd[p + "conv2.weight"], stride=stride, padding=1), new_stats, e)))
                o = _bn_train(sd, p + "bn3", conv(o, sd[p + "conv3.weight"]), new_stats, e)
                last = p + "conv3"
            else:
                o = tap(p + "conv1", F.relu(_bn_train(sd, p + "bn1", conv(x, sd[p + "conv1.weight"], stride=stride, padding=1), new_stats, e)))
                o = _bn_train(sd, p + "bn2", conv(o, sd[p + "conv2.weight"], padding=1), new_stats, e)
                last = p + "conv2"
            if p + "downsample.0.weight" in sd:
                identity = tap(p + "downsample.0", _bn_train(sd, p + "downsample.1", conv(x, sd[p + "downsample.0.weight"], stride=stride), new_stats, e))
            x = tap(last, _rnd(F.relu(o + identity), e))
    return F.adaptive_avg_pool2d(x, 1).flatten(1)


def resnet_encoder_train_step(sd: SD, model_name: str, x: torch.Tensor, dout: torch.Tensor, trainable,
                              emulate_bf16: bool = False, dtype=torch.float32):
    """Forward in training mode + backward of sum(out * dout).  `trainable`: names (with the `encoder.` prefix) that
    require a gradient.  Returns (out, {name: grad}, {running statistic name: new value})."""
    sd = {k: (v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for n in trainable:
        sd[n].requires_grad_(True)
    new_stats: Dict[str, torch.Tensor] = {}
    feat = resnet_trunk_train(sd, model_name, x.to(dtype), new_stats, emulate_bf16)
    out = F.relu(F.linear(feat, sd["encoder.embedding_layer.weight"], sd["encoder.embedding_layer.bias"]))
    (out * dout.to(dtype)).sum().backward()
    return out.detach(), {n: sd[n].grad.detach() for n in trainable}, {k: v.detach() for k, v in new_stats.items()}


def resnet_lstm_train_step(sd: SD, model_name: str, cfg: Dict, images: torch.Tensor, formulas: torch.Tensor, state: Dict,
                           trainable, lr: float = 1e-3, weight_decay: float = 1e-4, clip: float = 5.0, pad_id: int = 0) -> Dict:
    """One fp32 optimisation step of a resnet_lstm model (trainer.py:303-343, fp32 branch, dropout off): trunk in
    training mode (batch statistics, running statistics updated in ``sd``), Linear + ReLU (encoder.py:242-247), teacher-
    forced decoder, label-smoothed CE, clip by the global norm, Adam with coupled L2 over ``trainable`` only (torch's
    Adam skips parameters without a gradient: frozen ones get no update and no weight decay).  Updates ``sd`` in place."""
    import img2latex_oracle as O
    params = {k: (v.detach().clone().requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    new_stats: Dict[str, torch.Tensor] = {}
    feat = resnet_trunk_train(params, model_name, images, new_stats)
    enc = F.relu(F.linear(feat, params["encoder.embedding_layer.weight"], params["encoder.embedding_layer.bias"]))
    logits = O.decoder_forward(params, cfg, enc, formulas[:, :-1])
    loss = O.ce_label_smooth(logits, formulas[:, 1:], pad_id)
    names = [k for k in sd if k in trainable]
    gl = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    grads = {k: (torch.zeros_like(sd[k]) if g is None else g.detach().clone()) for k, g in zip(names, gl)}
    raw = {k: g.clone() for k, g in grads.items()}
    total = O.clip_grad_norm(grads, clip) if clip > 0 else torch.tensor(0.0)
    with torch.no_grad():
        O.adam_step({k: sd[k] for k in names}, grads, state, lr, weight_decay)
        for k, v in new_stats.items():
            sd[k].copy_(v)
    return dict(loss=float(loss.detach()), total_norm=float(total), grads=raw)
