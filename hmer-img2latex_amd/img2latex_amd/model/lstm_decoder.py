"""LSTMDecoder / Attention with the reference's surface, HIP step kernels.

Reference: img2latex/model/decoder.py:16-343.  nn.Embedding / nn.LSTM / nn.Linear
children only own the parameters under the reference's state_dict keys
(``embedding.weight``, ``lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{k}``,
``output_layer.{weight,bias}``, ``attention.attn.*``, ``attention.v.weight``); the
step math runs in libimg2latex_hip.so (csrc/decode.hip).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib

Hidden = Tuple[torch.Tensor, torch.Tensor]


class Attention(nn.Module):
    """Additive attention (decoder.py:287-343).  General source length via the HIP
    kernel; inside LSTMDecoder the source length is 1 and the context equals the
    encoder output bit-for-bit (softmax over one element), so the decoder never
    launches it (see LSTMDecoder docstring)."""

    def __init__(self, hidden_dim: int, encoder_dim: int):
        super().__init__()
        self.hidden_dim, self.encoder_dim = hidden_dim, encoder_dim
        self.attn = nn.Linear(hidden_dim + encoder_dim, hidden_dim)
        self.v = nn.Linear(hidden_dim, 1, bias=False)

    def forward(self, hidden: torch.Tensor, encoder_outputs: torch.Tensor) -> torch.Tensor:
        """hidden (B,1,H), encoder_outputs (B,S,E) -> context (B,1,E)."""
        hidden = _lib.require_gpu(hidden, "hidden")
        enc = _lib.require_gpu(encoder_outputs, "encoder_outputs")
        B, S, E = enc.shape
        if hidden.shape != (B, 1, self.hidden_dim) or E != self.encoder_dim:
            raise RuntimeError(f"Attention: bad shapes hidden {tuple(hidden.shape)} enc {tuple(enc.shape)}")
        ctx = torch.empty((B, 1, E), dtype=torch.float32, device=enc.device)
        _lib.check(_lib.lib().i2l_attention_context_fwd(
            hidden.data_ptr(), enc.data_ptr(), self.attn.weight.detach().contiguous().data_ptr(),
            self.attn.bias.detach().data_ptr(), self.v.weight.detach().contiguous().data_ptr(), ctx.data_ptr(),
            B, S, self.hidden_dim, E, _lib.stream_ptr()), "attention_context_fwd")
        return ctx


class LSTMDecoder(nn.Module):
    """decoder.py:16-284.

    Attention note: the encoder emits ONE vector per image, so Attention runs over a
    source of length 1: softmax over one element is exactly 1.0 and
    ``context == encoder_output`` bit-for-bit (decoder.py:338-341; golden fixture G7).
    The step kernels therefore consume ``encoder_output`` directly for both
    ``attention=True`` and ``attention=False``; results are identical to the reference's.
    """

    def __init__(self, vocab_size: int, embedding_dim: int = None, hidden_dim: int = None,
                 max_seq_length: int = None, lstm_layers: int = None, dropout: float = None,
                 attention: bool = True):
        super().__init__()
        # defaults of decoder.py:49-58
        embedding_dim = 256 if embedding_dim is None else embedding_dim
        hidden_dim = 256 if hidden_dim is None else hidden_dim
        max_seq_length = 141 if max_seq_length is None else max_seq_length
        lstm_layers = 1 if lstm_layers is None else lstm_layers
        dropout = 0.1 if dropout is None else dropout
        self.vocab_size, self.embedding_dim, self.hidden_dim = vocab_size, embedding_dim, hidden_dim
        self.max_seq_length, self.lstm_layers, self.dropout = max_seq_length, lstm_layers, dropout
        self.use_attention = attention
        # creation order = the reference's (decoder.py:69-93) so torch.manual_seed gives the same init
        self.embedding = nn.Embedding(vocab_size, embedding_dim)
        self.lstm = nn.LSTM(input_size=2 * embedding_dim, hidden_size=hidden_dim, num_layers=lstm_layers,
                            batch_first=True, dropout=dropout if lstm_layers > 1 else 0)
        if attention:
            self.attention = Attention(hidden_dim, embedding_dim)
        self.output_layer = nn.Linear(hidden_dim, vocab_size)
        self.dropout_layer = nn.Dropout(dropout)
        self._ws: Optional[torch.Tensor] = None      # workspace of the CURRENT stream (see _select_workspace)
        self._ws_key = None          # (rows, weight versions) the weight images in _ws were built for
        self._ws_by_stream = {}      # stream handle -> (workspace, key): concurrent decodes must not share one
        self.kernel_flags = 0        # _lib.FLAG_NO_GROUP / FLAG_EXACT_FP32 (training), FLAG_AGENT_SCOPE_EXCHANGE (all grouped)

    # ------------------------------------------------------------------ plumbing
    def _weights_struct(self):
        """(struct, keep-alive list) describing the parameters for the C ABI."""
        L = self.lstm_layers
        names = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
        tens = {n: [_lib.require_gpu(getattr(self.lstm, f"{n}_l{l}").detach(), f"lstm.{n}_l{l}") for l in range(L)]
                for n in names}
        emb = _lib.require_gpu(self.embedding.weight.detach(), "embedding.weight")
        wout = _lib.require_gpu(self.output_layer.weight.detach(), "output_layer.weight")
        bout = _lib.require_gpu(self.output_layer.bias.detach(), "output_layer.bias")
        arrs = {n: _lib.pointer_array(tens[n]) for n in names}
        w = _lib.DecoderWeights()
        w.embedding = emb.data_ptr()
        w.w_ih, w.w_hh = arrs["weight_ih"], arrs["weight_hh"]
        w.b_ih, w.b_hh = arrs["bias_ih"], arrs["bias_hh"]
        w.w_out, w.b_out = wout.data_ptr(), bout.data_ptr()
        w.vocab, w.embed, w.hidden, w.layers = self.vocab_size, self.embedding_dim, self.hidden_dim, L
        return w, (tens, emb, wout, bout, arrs)

    def _weights_version(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def prepare(self, encoder_output: torch.Tensor, reuse_weight_images: bool = True, slot=None):
        """Build the decode workspace for these encoder rows (i2l_decoder_prepare).
        ``slot`` (any hashable): use the workspace of that name instead of the current stream's -- a pipeline prepares batch
        i + 1 on its encoder stream while decode(i) still reads the workspace of batch i, then passes what this returns
        (plus ``self._ws``) to ``run_steps(prepared=...)``.
        The weight images (transposed / gate-interleaved matrices, the token table P) are rebuilt only
        when a parameter changed since they were built -- (data_ptr, _version) of every parameter is
        the cache key, the same contract as nn.LSTM's flattened weights; code that writes parameters
        through raw pointers must bump the version (TrainStep.apply does).  ``reuse_weight_images=False``
        forces the rebuild.  The image-dependent rows (Genc) are rebuilt on every call."""
        enc = _lib.require_gpu(encoder_output, "encoder_output")
        if enc.dim() != 2 or enc.shape[1] != self.embedding_dim:
            raise RuntimeError(f"encoder_output must be (B,{self.embedding_dim}), got {tuple(enc.shape)}")
        rows = enc.shape[0]
        L = _lib.lib()
        w, keep = self._weights_struct()
        nbytes = L.i2l_decoder_workspace_bytes(rows, self.vocab_size, self.embedding_dim, self.hidden_dim,
                                               self.lstm_layers)
        if nbytes == 0:
            raise RuntimeError("img2latex_amd: decoder dimensions not supported by the HIP kernels")
        key = (rows, enc.device, self._weights_version())
        what = _lib.PREP_ALL
        sid = _lib.stream_ptr() if slot is None else ("slot", slot)
        self._ws, self._ws_key = self._ws_by_stream.get(sid, (None, None))
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != enc.device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=enc.device)
            self._ws_key = None
        if reuse_weight_images and self._ws_key == key:
            what = _lib.PREP_ROWS
        _lib.check(L.i2l_decoder_prepare(ctypes.byref(w), enc.data_ptr(), rows, what, self._ws.data_ptr(),
                                         self._ws.numel(), _lib.stream_ptr()), "decoder_prepare")
        self._ws_key = key
        self._ws_by_stream[sid] = (self._ws, key)
        _lib.mark("prepare")
        return w, keep, enc

    def release_slots(self, owner) -> None:
        """Drop the workspaces ``prepare(slot=(..., owner, ...))`` created for ``owner`` (a pipeline's id): a pipeline that
        goes away gives its decoder workspaces back instead of leaving them pinned on the model."""
        for sid in [k for k in self._ws_by_stream
                    if isinstance(k, tuple) and k[0] == "slot" and isinstance(k[1], tuple) and len(k[1]) > 1 and k[1][1] == owner]:
            ws, _ = self._ws_by_stream.pop(sid)
            if self._ws is ws:
                self._ws, self._ws_key = None, None

    def group_status(self):
        """Diagnostics of the LAST grouped greedy launch on the current stream (synchronises): {"timed_out", "groups",
        "groups_on_one_xcd"} -- how many 4-workgroup groups ran and how many of them found their members on one XCD
        and exchanged through its L2 -- or None when these dimensions / this batch have no grouped path."""
        if self._ws is None or self._ws_key is None:
            return None
        off = _lib.lib().i2l_decoder_group_status_offset(self._ws_key[0], self.vocab_size, self.embedding_dim,
                                                         self.hidden_dim, self.lstm_layers)
        if off == 0:
            return None
        st = self._ws[off:off + 16].view(torch.int32).cpu().tolist()
        return {"timed_out": bool(st[0]), "groups": st[1], "groups_on_one_xcd": st[2]}

    def run_steps(self, encoder_output: torch.Tensor, steps: int, tok0: torch.Tensor,
                  forced: Optional[torch.Tensor] = None, hidden: Optional[Hidden] = None,
                  temperature: float = 1.0, select: int = _lib.SELECT_LOGITS, stop: int = _lib.STOP_NONE,
                  end_id: int = -1, want_ids: bool = True, want_logits: bool = False,
                  want_state: bool = False, reuse_weight_images: bool = True, rows_per_workgroup: int = 0,
                  flags: int = 0, prepared=None, resident=None):
        """prepare + one persistent i2l_greedy_decode launch.  Returns (ids, logits, (h, c)).
        ``prepared`` = (w, keep, enc, workspace) of an earlier ``prepare(encoder_output, slot=...)`` whose completion the
        current stream already waits for: the launch uses that workspace and prepares nothing.
        ``resident`` = (int32 device tensor, value): the launch stores ``value`` there once its workgroups own their compute
        units (i2l_greedy_decode_ex's residency signal; GreedyPipeline's encoder stream waits for it)."""
        if prepared is None:
            w, keep, enc = self.prepare(encoder_output, reuse_weight_images)
            ws = self._ws
        else:
            w, keep, enc, ws = prepared
        rows, dev = enc.shape[0], enc.device
        tok0 = _lib.require_gpu(tok0, "tok0", torch.int32)
        if forced is not None:
            forced = _lib.require_gpu(forced, "forced", torch.int32)
            if forced.shape != (rows, steps):
                raise RuntimeError(f"forced tokens must be ({rows},{steps}), got {tuple(forced.shape)}")
        h0 = c0 = None
        if hidden is not None:
            h0 = _lib.require_gpu(hidden[0], "h0")
            c0 = _lib.require_gpu(hidden[1], "c0")
            shp = (self.lstm_layers, rows, self.hidden_dim)
            if tuple(h0.shape) != shp or tuple(c0.shape) != shp:
                raise RuntimeError(f"hidden state must be {shp}, got {tuple(h0.shape)} / {tuple(c0.shape)}")
        ids = torch.empty((rows, steps), dtype=torch.int32, device=dev) if want_ids else None
        logits = torch.empty((rows, steps, self.vocab_size), dtype=torch.float32, device=dev) if want_logits else None
        h = c = None
        if want_state:
            h = torch.empty((self.lstm_layers, rows, self.hidden_dim), dtype=torch.float32, device=dev)
            c = torch.empty_like(h)
        _lib.check(_lib.lib().i2l_greedy_decode_ex(
            ctypes.byref(w), ws.data_ptr(), rows, steps, tok0.data_ptr(), _lib.ptr(forced), _lib.ptr(h0),
            _lib.ptr(c0), float(temperature), select, stop, int(end_id), int(rows_per_workgroup), _lib.ptr(ids),
            _lib.ptr(logits), _lib.ptr(h), _lib.ptr(c), int(flags) | int(self.kernel_flags),
            None if resident is None else resident[0].data_ptr(), 0 if resident is None else int(resident[1]) & 0xFFFFFFFF,
            _lib.stream_ptr()), "greedy_decode")
        _lib.mark("decode")
        del keep
        return ids, logits, ((h, c) if want_state else None)

    def sample_steps(self, encoder_output: torch.Tensor, steps: int, tok0: torch.Tensor, temperature: float,
                     top_k: int, top_p: float, seed: int, stop: int = _lib.STOP_STICKY, end_id: int = -1,
                     hidden: Optional[Hidden] = None, want_probs: bool = False):
        """prepare + one i2l_sample_decode launch (predictor.py:295-331).  Returns (ids, probs or None)."""
        w, keep, enc = self.prepare(encoder_output)
        rows, dev = enc.shape[0], enc.device
        tok0 = _lib.require_gpu(tok0, "tok0", torch.int32)
        h0 = c0 = None
        if hidden is not None:
            h0, c0 = _lib.require_gpu(hidden[0], "h0"), _lib.require_gpu(hidden[1], "c0")
        ids = torch.empty((rows, steps), dtype=torch.int32, device=dev)
        probs = torch.empty((rows, steps, self.vocab_size), dtype=torch.float32, device=dev) if want_probs else None
        _lib.check(_lib.lib().i2l_sample_decode(
            ctypes.byref(w), self._ws.data_ptr(), rows, steps, tok0.data_ptr(), _lib.ptr(h0), _lib.ptr(c0),
            float(temperature), int(top_k), float(top_p), int(seed) & 0xFFFFFFFFFFFFFFFF, stop, int(end_id),
            ids.data_ptr(), _lib.ptr(probs), None, None, _lib.stream_ptr()), "sample_decode")
        del keep
        return ids, probs

    # ------------------------------------------------------------------ reference surface
    def forward(self, encoder_output: torch.Tensor, target_sequence: torch.Tensor, hidden=None) -> torch.Tensor:
        """Teacher forcing (decoder.py:100-195): (B,E), (B,T) int64 -> logits (B,T,V).
        With gradients enabled (or dropout active) the training kernels run and the result
        carries a grad_fn; otherwise the persistent decode kernel runs with forced tokens."""
        tgt = target_sequence
        if not tgt.is_cuda:
            raise RuntimeError("img2latex_amd: target_sequence must be on the ROCm device (no CPU fallback)")
        B, T = tgt.shape
        forced = tgt.to(torch.int32).contiguous()
        needs_grad = torch.is_grad_enabled() and (encoder_output.requires_grad
                                                  or any(p.requires_grad for p in self.parameters()))
        if needs_grad or (self.training and self.dropout > 0):
            # training-mode forward: keeps the BPTT state; dropout masks are drawn from a fresh seed
            if hidden is not None:
                raise NotImplementedError("img2latex_amd: training forward starts from a zero state (as the Trainer does)")
            from ._train_fn import DecoderTeacherForcedFn
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            return DecoderTeacherForcedFn.apply(self, encoder_output, forced, seed, *self.parameters())
        _, logits, _ = self.run_steps(encoder_output, T, forced[:, 0].contiguous(), forced=forced, hidden=hidden,
                                      want_ids=False, want_logits=True)
        return logits

    def decode_step(self, encoder_output: torch.Tensor, input_token: torch.Tensor, hidden=None
                    ) -> Tuple[torch.Tensor, Hidden]:
        """One step (decoder.py:197-284): (B,E), (B,1) int64, (h,c)|None -> (logits (B,1,V), (h,c)).
        Returns fresh state tensors and never mutates its inputs (seq2seq.py:272 relies on it)."""
        if input_token.dim() != 2 or input_token.shape[1] != 1:
            raise RuntimeError(f"Shape mismatch: input_token must be (B,1), got {tuple(input_token.shape)}")
        if not input_token.is_cuda:
            raise RuntimeError("img2latex_amd: input_token must be on the ROCm device (no CPU fallback)")
        tok0 = input_token.reshape(-1).to(torch.int32).contiguous()
        _, logits, state = self.run_steps(encoder_output, 1, tok0, hidden=hidden, want_ids=False,
                                          want_logits=True, want_state=True, reuse_weight_images=True)
        return logits, state
