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
        enc = _lib.require_gpu(enc.detach(), "encoder_output")
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
                                           logits.data_ptr(), _lib.stream_ptr()), "decoder_train_fwd")
        del keep
        ctx.decoder, ctx.ws, ctx.tokens, ctx.seed, ctx.p = decoder, ws, tokens, int(seed), p
        ctx.enc_shape = enc.shape
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        dec = ctx.decoder
        dlogits = _lib.require_gpu(dlogits, "dlogits")
        B, T = ctx.tokens.shape
        L = _lib.lib()
        w, keep = dec._weights_struct()
        named = list(dec.named_parameters())
        grads = {n: torch.zeros_like(p_) if n.startswith("attention.") else torch.empty_like(p_) for n, p_ in named}
        Ln = dec.lstm_layers
        arrs = {k: _lib.pointer_array([grads[f"lstm.{k}_l{l}"] for l in range(Ln)])
                for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
        g = _lib.DecoderGrads()
        g.embedding = grads["embedding.weight"].data_ptr()
        g.w_ih, g.w_hh, g.b_ih, g.b_hh = arrs["weight_ih"], arrs["weight_hh"], arrs["bias_ih"], arrs["bias_hh"]
        g.w_out, g.b_out = grads["output_layer.weight"].data_ptr(), grads["output_layer.bias"].data_ptr()
        denc = torch.empty(ctx.enc_shape, dtype=torch.float32, device=dlogits.device)
        _lib.check(L.i2l_decoder_train_bwd(ctypes.byref(w), ctx.tokens.data_ptr(), B, T, ctx.p, ctx.seed,
                                           1 if dec.use_attention else 0, ctx.ws.data_ptr(), ctx.ws.numel(),
                                           dlogits.data_ptr(), ctypes.byref(g), denc.data_ptr(), _lib.stream_ptr()),
                   "decoder_train_bwd")
        del keep
        ctx.ws = None
        return (None, denc, None, None) + tuple(grads[n] for n, _ in named)


class CNNEncoderFn(torch.autograd.Function):
    """CNNEncoder.forward with a HIP backward (autograd of encoder.py:111-129).

    inputs: encoder module, images, then the parameters in ``encoder.parameters()`` order
    (conv weight/bias per block, embedding_layer weight/bias)."""

    @staticmethod
    def forward(ctx, encoder, x, *params):
        x = _lib.require_gpu(x.detach(), "images")
        amax, blocks = [], []
        out = encoder._forward_impl(x, argmax_out=amax, blocks_out=blocks)
        ctx.encoder, ctx.x, ctx.blocks, ctx.amax, ctx.out = encoder, x, blocks, amax, out
        ctx.need_dx = False
        return out

    @staticmethod
    def backward(ctx, denc):
        enc = ctx.encoder
        L = _lib.lib()
        denc = _lib.require_gpu(denc, "d encoder_output")
        dev = denc.device
        named = list(enc.named_parameters())
        grads = {n: torch.empty_like(p_) for n, p_ in named}
        feat = ctx.blocks[-1]
        B = feat.shape[0]
        K, E = feat.numel() // B, enc.embedding_dim
        lin = enc.embedding_layer
        dfeat = torch.empty_like(feat)
        nbytes = L.i2l_linear_bwd_workspace_bytes(B, K, E)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(L.i2l_linear_bias_act_bwd(feat.data_ptr(), lin.weight.detach().data_ptr(), ctx.out.data_ptr(),
                                             denc.data_ptr(), dfeat.data_ptr(), grads["embedding_layer.weight"].data_ptr(),
                                             grads["embedding_layer.bias"].data_ptr(), B, K, E, 1, ws.data_ptr(), nbytes,
                                             _lib.stream_ptr()), "linear_bias_act_bwd")
        dy = dfeat
        nblk = len(enc.conv_filters)
        for i in reversed(range(nblk)):
            conv = enc.cnn_layers[3 * i]
            xin = ctx.x if i == 0 else ctx.blocks[i - 1]
            _, cin, h, w = xin.shape
            dx = torch.empty_like(xin) if i > 0 else None          # images need no gradient
            nbytes = L.i2l_conv_bwd_workspace_bytes(B, cin, h, w, conv.out_channels)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.i2l_conv3x3_relu_pool2_bwd(
                xin.data_ptr(), conv.weight.detach().data_ptr(), ctx.blocks[i].data_ptr(), ctx.amax[i].data_ptr(),
                dy.data_ptr(), _lib.ptr(dx), grads[f"cnn_layers.{3 * i}.weight"].data_ptr(),
                grads[f"cnn_layers.{3 * i}.bias"].data_ptr(), B, cin, h, w, conv.out_channels, ws.data_ptr(), nbytes,
                _lib.stream_ptr()), "conv3x3_relu_pool2_bwd")
            dy = dx
        ctx.blocks = ctx.amax = None
        return (None, None) + tuple(grads[n] for n, _ in named)
