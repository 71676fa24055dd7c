"""torch.autograd bridges for the training-mode forward/backward HIP kernels.

They let the reference's Trainer code (`loss = criterion(model(images, formulas).transpose(1,2),
targets); loss.backward()`, trainer.py:334-337) run unchanged on the HIP path: the forward
kernels keep what the backward kernels need in a workspace tensor held by the autograd node.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _lib


class DecoderTeacherForcedFn(torch.autograd.Function):
    """LSTMDecoder.forward in training mode (decoder.py:100-195).

    inputs: decoder module, enc (B,E), tokens (B,T) int32, seed, then the parameters in
    ``decoder.parameters()`` order.  Attention parameters (if any) get exactly-zero gradients,
    as in the reference (the context over a length-1 source does not depend on them)."""

    @staticmethod
    def forward(ctx, decoder, enc, tokens, seed, *params):
        logits, state = decoder_train_forward(decoder, enc.detach(), tokens, seed)
        ctx.decoder, ctx.state = decoder, state
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        dec = ctx.decoder
        named = list(dec.named_parameters())
        grads = {n: torch.empty_like(p_) for n, p_ in named}
        denc = decoder_train_backward(dec, ctx.state, dlogits, grads)
        ctx.state = None
        return (None, denc, None, None) + tuple(grads[n] for n, _ in named)


def decoder_train_forward(decoder, enc, tokens, seed):
    """i2l_decoder_train_fwd: returns (logits (B,T,V), saved state for decoder_train_backward)."""
    enc = _lib.require_gpu(enc, "encoder_output")
    tokens = _lib.require_gpu(tokens, "tokens", torch.int32)
    B, T = tokens.shape
    L = _lib.lib()
    w, keep = decoder._weights_struct()
    nbytes = L.i2l_decoder_train_workspace_bytes(B, T, decoder.vocab_size, decoder.embedding_dim,
                                                 decoder.hidden_dim, decoder.lstm_layers)
    if nbytes == 0:
        raise RuntimeError("img2latex_amd: decoder dimensions not supported by the HIP training kernels")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=enc.device)
    logits = torch.empty((B, T, decoder.vocab_size), dtype=torch.float32, device=enc.device)
    p = float(decoder.dropout) if decoder.training else 0.0
    _lib.check(L.i2l_decoder_train_fwd(ctypes.byref(w), enc.data_ptr(), tokens.data_ptr(), B, T, p, int(seed),
                                       1 if decoder.use_attention else 0, ws.data_ptr(), nbytes,
                                       logits.data_ptr(), int(decoder.kernel_flags), _lib.stream_ptr()),
               "decoder_train_fwd")
    del keep
    return logits, dict(ws=ws, tokens=tokens, seed=int(seed), p=p, enc_shape=tuple(enc.shape),
                        flags=int(decoder.kernel_flags))


def decoder_train_backward(dec, state, dlogits, grads, lanes=None):
    """i2l_decoder_train_bwd: fills ``grads`` (name -> tensor, ``decoder.named_parameters()`` names;
    attention parameters are zero-filled) and returns d(encoder_output).  ``lanes``: an ``_lib.Lanes(...).handle`` puts the
    weight gradients on the caller's side lanes -- the caller keeps ``state`` / ``dlogits`` alive and joins
    (``Lanes.join``) before reading ``grads`` (TrainStep does)."""
    dlogits = _lib.require_gpu(dlogits, "dlogits")
    tokens = state["tokens"]
    B, T = tokens.shape
    L = _lib.lib()
    w, keep = dec._weights_struct()
    for n, t in grads.items():
        if n.startswith("attention."):
            t.zero_()
    Ln = dec.lstm_layers
    arrs = {k: _lib.pointer_array([grads[f"lstm.{k}_l{l}"] for l in range(Ln)])
            for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
    g = _lib.DecoderGrads()
    g.embedding = grads["embedding.weight"].data_ptr()
    g.w_ih, g.w_hh, g.b_ih, g.b_hh = arrs["weight_ih"], arrs["weight_hh"], arrs["bias_ih"], arrs["bias_hh"]
    g.w_out, g.b_out = grads["output_layer.weight"].data_ptr(), grads["output_layer.bias"].data_ptr()
    denc = torch.empty(state["enc_shape"], dtype=torch.float32, device=dlogits.device)
    ws = state["ws"]
    _lib.check(L.i2l_decoder_train_bwd(ctypes.byref(w), tokens.data_ptr(), B, T, state["p"], state["seed"],
                                       1 if dec.use_attention else 0, ws.data_ptr(), ws.numel(),
                                       dlogits.data_ptr(), ctypes.byref(g), denc.data_ptr(),
                                       state["flags"], lanes, _lib.stream_ptr()),
               "decoder_train_bwd")
    del keep
    return denc


class CNNEncoderFn(torch.autograd.Function):
    """CNNEncoder.forward with a HIP backward (autograd of encoder.py:111-129).

    inputs: encoder module, images, then the parameters in ``encoder.parameters()`` order
    (conv weight/bias per block, embedding_layer weight/bias)."""

    @staticmethod
    def forward(ctx, encoder, x, *params):
        out, state = encoder_train_forward(encoder, x.detach())
        ctx.encoder, ctx.state = encoder, state
        return out

    @staticmethod
    def backward(ctx, denc):
        enc = ctx.encoder
        named = list(enc.named_parameters())
        grads = {n: torch.empty_like(p_) for n, p_ in named}
        encoder_train_backward(enc, ctx.state, denc, grads)
        ctx.state = None
        return (None, None) + tuple(grads[n] for n, _ in named)


class ResNetEncoderFn(torch.autograd.Function):
    """ResNetEncoder.forward in training mode (encoder.py:231-249 under model.train()) with a HIP backward.
    inputs: encoder module, images, then the TRAINABLE parameters in ``encoder.parameters()`` order."""

    @staticmethod
    def forward(ctx, encoder, x, *params):
        out, state = encoder_train_forward(encoder, x.detach())
        ctx.encoder, ctx.state = encoder, state
        return out

    @staticmethod
    def backward(ctx, denc):
        enc = ctx.encoder
        named = [(n, p_) for n, p_ in enc.named_parameters() if p_.requires_grad]
        grads = {n: torch.zeros_like(p_) for n, p_ in named}
        encoder_train_backward(enc, ctx.state, denc.contiguous(), grads)
        ctx.state = None
        return (None, None) + tuple(grads[n] for n, _ in named)


def _resnet_train_forward(encoder, x):
    tape = {}
    feat = encoder._trunk_train(x, tape)
    out = encoder._head(feat)
    tape.update(feat=feat, out=out)
    return out, tape


def _resnet_train_backward(enc, tape, denc, grads, after_linear=None):
    """``grads``: name -> tensor for the TRAINABLE parameters of the encoder (``named_parameters()`` names)."""
    L = _lib.lib()
    denc = _lib.require_gpu(denc, "d encoder_output")
    feat, out = tape["feat"], tape["out"]
    B, K = feat.shape
    E = enc.embedding_dim
    lin = enc.embedding_layer
    dev = denc.device
    dfeat = torch.empty_like(feat)
    dw = grads.get("embedding_layer.weight")
    db = grads.get("embedding_layer.bias")
    dw = dw if dw is not None else torch.empty_like(lin.weight)
    db = db if db is not None else torch.empty_like(lin.bias)
    nbytes = L.i2l_linear_bwd_workspace_bytes(B, K, E)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(L.i2l_linear_bias_act_bwd(feat.data_ptr(), lin.weight.detach().data_ptr(), out.data_ptr(), denc.data_ptr(),
                                         dfeat.data_ptr(), dw.data_ptr(), db.data_ptr(), B, K, E, 1, ws.data_ptr(), nbytes,
                                         0, None, _lib.stream_ptr()), "linear_bias_act_bwd")
    if after_linear is not None:
        after_linear()
    enc._trunk_backward(tape, dfeat, grads)


def encoder_train_forward(encoder, x):
    x = _lib.require_gpu(x, "images")
    if hasattr(encoder, "_trunk_train"):                  # ResNetEncoder
        return _resnet_train_forward(encoder, x)
    amax, blocks = [], []
    out = encoder._forward_impl(x, argmax_out=amax, blocks_out=blocks)
    return out, dict(x=x, blocks=blocks, amax=amax, out=out)


def encoder_train_backward(enc, state, denc, grads, after_linear=None, lanes=None, keep=None):
    """Fills ``grads`` (``encoder.named_parameters()`` names) from d(encoder_output).  ``after_linear`` is called once the
    FC layer's backward has been enqueued (TrainStep starts the early part of the gradient all-reduce there).
    ``lanes`` = an ``_lib.Lanes(...).handle`` (CNN encoder): weight gradients on the caller's side lanes; every temporary
    they read is appended to the list ``keep``, which the caller holds until it has joined (``Lanes.join``)."""
    if hasattr(enc, "_trunk_train"):                      # ResNetEncoder (its gradient GEMMs stay on the caller's stream)
        return _resnet_train_backward(enc, state, denc, grads, after_linear)
    side = lanes
    if side and keep is None:
        raise ValueError("side lanes need a `keep` list that outlives the join")
    L = _lib.lib()
    denc = _lib.require_gpu(denc, "d encoder_output")
    dev = denc.device
    blocks, amax = state["blocks"], state["amax"]
    feat = blocks[-1]
    B = feat.shape[0]
    K, E = feat.numel() // B, enc.embedding_dim
    lin = enc.embedding_layer
    dfeat = torch.empty_like(feat)
    nbytes = L.i2l_linear_bwd_workspace_bytes(B, K, E)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(L.i2l_linear_bias_act_bwd(feat.data_ptr(), lin.weight.detach().data_ptr(), state["out"].data_ptr(),
                                         denc.data_ptr(), dfeat.data_ptr(), grads["embedding_layer.weight"].data_ptr(),
                                         grads["embedding_layer.bias"].data_ptr(), B, K, E, 1, ws.data_ptr(), nbytes,
                                         enc.kernel_flags, side, _lib.stream_ptr()), "linear_bias_act_bwd")
    if side:
        keep.extend((ws, denc, state))
    if after_linear is not None:
        after_linear()
    dy = dfeat
    for i in reversed(range(len(enc.conv_filters))):
        conv = enc.cnn_layers[3 * i]
        xin = state["x"] if i == 0 else blocks[i - 1]
        _, cin, h, w = xin.shape
        dx = torch.empty_like(xin) if i > 0 else None          # images need no gradient
        nbytes = L.i2l_conv_bwd_workspace_bytes(B, cin, h, w, conv.out_channels)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(L.i2l_conv3x3_relu_pool2_bwd(
            xin.data_ptr(), conv.weight.detach().data_ptr(), blocks[i].data_ptr(), amax[i].data_ptr(),
            dy.data_ptr(), _lib.ptr(dx), grads[f"cnn_layers.{3 * i}.weight"].data_ptr(),
            grads[f"cnn_layers.{3 * i}.bias"].data_ptr(), B, cin, h, w, conv.out_channels, ws.data_ptr(), nbytes,
            enc.kernel_flags, side, _lib.stream_ptr()), "conv3x3_relu_pool2_bwd")
        if side:
            keep.extend((ws, dy))
        dy = dx
