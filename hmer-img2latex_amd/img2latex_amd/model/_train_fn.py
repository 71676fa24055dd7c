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
