"""Seq2SeqModel with the reference's surface (img2latex/model/seq2seq.py:17-298).

``forward`` / ``inference`` / ``_greedy_search`` / ``_beam_search`` keep the reference's
argument meaning, defaults, return types and post-processing; the encoder and the
whole decode loop run as HIP kernels (one persistent launch per search, one
device->host copy of the ids at the end instead of a sync per step).
"""
from __future__ import annotations

import ctypes
import warnings
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .cnn_encoder import CNNEncoder
from .lstm_decoder import LSTMDecoder


class Seq2SeqModel(nn.Module):
    def __init__(self, model_type: str = "cnn_lstm", vocab_size: int = None,
                 encoder_params: Dict = None, decoder_params: Dict = None):
        super().__init__()
        encoder_params = {} if encoder_params is None else encoder_params
        decoder_params = {} if decoder_params is None else decoder_params
        vocab_size = 100 if vocab_size is None else vocab_size              # seq2seq.py:50-51
        embedding_dim = encoder_params.get("embedding_dim", 256)           # :54
        if model_type == "cnn_lstm":                                        # defaults of :58-67
            self.encoder = CNNEncoder(
                img_height=encoder_params.get("img_height", 50), img_width=encoder_params.get("img_width", 200),
                channels=encoder_params.get("channels", 1),
                conv_filters=encoder_params.get("conv_filters", [32, 64, 128]),
                kernel_size=encoder_params.get("kernel_size", 3), pool_size=encoder_params.get("pool_size", 2),
                padding=encoder_params.get("padding", "same"), embedding_dim=embedding_dim)
        elif model_type == "resnet_lstm":
            from .resnet_encoder import ResNetEncoder                       # defaults of :69-76
            self.encoder = ResNetEncoder(
                img_height=encoder_params.get("img_height", 224), img_width=encoder_params.get("img_width", 224),
                channels=encoder_params.get("channels", 3), model_name=encoder_params.get("model_name", "resnet50"),
                embedding_dim=embedding_dim, freeze_backbone=encoder_params.get("freeze_backbone", True))
        else:
            raise ValueError(f"Invalid model type: {model_type}. Expected 'cnn_lstm' or 'resnet_lstm'.")
        self.decoder = LSTMDecoder(                                         # defaults of :83-91
            vocab_size=vocab_size, embedding_dim=embedding_dim, hidden_dim=decoder_params.get("hidden_dim", 256),
            max_seq_length=decoder_params.get("max_seq_length", 150), lstm_layers=decoder_params.get("lstm_layers", 1),
            dropout=decoder_params.get("dropout", 0.1), attention=decoder_params.get("attention", False))
        self.model_type = model_type
        self.vocab_size = vocab_size

    # ---------------------------------------------------------------- training-mode forward
    def forward(self, images: torch.Tensor, target_sequences: torch.Tensor) -> torch.Tensor:
        """decoder(encoder(images), target_sequences[:, :-1]) -> (B,T,V)   (seq2seq.py:98-122)."""
        return self.decoder(self.encoder(images), target_sequences[:, :-1])

    # ---------------------------------------------------------------- inference
    def inference(self, image: torch.Tensor, start_token_id: int, end_token_id: int, max_length: int = None,
                  temperature: float = None, top_k: int = None, top_p: float = None, beam_size: int = None):
        """seq2seq.py:124-190: List[int] for one image, List[List[int]] for a batch."""
        max_length = 150 if max_length is None else max_length
        temperature = 1.0 if temperature is None else temperature
        top_k = 0 if top_k is None else top_k
        top_p = 0.0 if top_p is None else top_p
        beam_size = 0 if beam_size is None else beam_size
        encoder_output = self.encoder(image)
        if encoder_output.dim() == 1:
            encoder_output = encoder_output.unsqueeze(0)
        if beam_size > 0:
            return self._beam_search(encoder_output, start_token_id, end_token_id, max_length, beam_size)
        return self._greedy_search(encoder_output, start_token_id, end_token_id, max_length, temperature, top_k, top_p)

    def greedy_ids(self, encoder_output: torch.Tensor, start_token_id: int, end_token_id: int, max_length: int,
                   temperature: float = 1.0, stop: int = _lib.STOP_NONE, select: int = _lib.SELECT_LOGITS,
                   want_logits: bool = False, rows_per_workgroup: int = 0, flags: int = 0, prepared=None, resident=None):
        """Device-side greedy loop; returns (ids (B,T) int32 on device, logits or None).  ``flags``:
        _lib.FLAG_DECODE_GROUP8 / _GROUP16 select the 8-member (vector ALUs) / 16-member (matrix cores) grouped kernels (the
        ones that share a CU with a conv workgroup);
        ``prepared`` / ``resident``: see LSTMDecoder.run_steps."""
        B = encoder_output.shape[0]
        key = (B, int(start_token_id), encoder_output.device)
        if getattr(self, "_tok0_key", None) != key:               # the START column: built once per batch shape
            self._tok0 = torch.full((B,), int(start_token_id), dtype=torch.int32, device=encoder_output.device)
            self._tok0_key = key
        tok0 = self._tok0
        ids, logits, _ = self.decoder.run_steps(encoder_output, max_length, tok0, temperature=temperature,
                                                select=select, stop=stop, end_id=end_token_id,
                                                want_logits=want_logits, rows_per_workgroup=rows_per_workgroup,
                                                flags=flags, prepared=prepared, resident=resident)
        return ids, logits

    def greedy_ids_host(self, encoder_output: torch.Tensor, start_token_id: int, end_token_id: int, max_length: int,
                        temperature: float = 1.0, stop: int = _lib.STOP_NONE, select: int = _lib.SELECT_LOGITS
                        ) -> torch.Tensor:
        """greedy_ids + the ONE device->host copy of the search, with the timeout fallback both callers
        (_greedy_search and Predictor.predict_batch_ids) share: the grouped kernel needs its 4 members resident
        together; on a GPU shared with other work (fewer than 32 free CUs) a bounded wait can expire (ids -3) ->
        run the row-per-workgroup HIP kernel, which needs no partner."""
        ids, _ = self.greedy_ids(encoder_output, start_token_id, end_token_id, max_length, temperature, stop=stop,
                                 select=select)
        ids = ids.cpu()
        if _lib.ids_timed_out(ids):
            warnings.warn("img2latex_amd: grouped decode timed out (GPU oversubscribed?); "
                          "re-running on the row-per-workgroup kernel", RuntimeWarning)
            ids, _ = self.greedy_ids(encoder_output, start_token_id, end_token_id, max_length, temperature, stop=stop,
                                     select=select, rows_per_workgroup=1)
            ids = _lib.check_ids(ids.cpu())
        return ids

    def _greedy_search(self, encoder_output: torch.Tensor, start_token_id: int, end_token_id: int,
                       max_length: int, temperature: float, top_k: int, top_p: float):
        """seq2seq.py:192-232.  top_k / top_p are accepted and ignored, as in the reference.
        The loop stops when ALL rows emit END in the same step (:220); the kernel runs the
        rows independently, so that step is located in the ids afterwards."""
        B = encoder_output.shape[0]
        ids = self.greedy_ids_host(encoder_output, start_token_id, end_token_id, max_length, temperature).numpy()
        # host post-processing in numpy: one thread, no dispatch (torch would fan a 38 k-element compare out to every
        # host core: measured 10 ms per batch on the 16-core GPU box against 1.4 ms for the whole device path)
        all_end = (ids == end_token_id).all(axis=0)
        steps = int(all_end.argmax()) + 1 if bool(all_end.any()) else max_length
        full = np.empty((B, steps + 1), dtype=np.int64)
        full[:, 0] = start_token_id
        full[:, 1:] = ids[:, :steps]
        sequences = full.tolist()
        seq = sequences[0] if B == 1 else sequences
        if B == 1:                                                # :224-231
            if seq and seq[0] == start_token_id:
                seq = seq[1:]
            if end_token_id in seq:
                seq = seq[: seq.index(end_token_id)]
        return seq

    def _beam_search(self, encoder_output: torch.Tensor, start_token_id: int, end_token_id: int,
                     max_length: int, beam_size: int):
        """seq2seq.py:234-298: batch size 1 only; a batch falls back to greedy (:244-247)."""
        if encoder_output.size(0) != 1:
            return self._greedy_search(encoder_output, start_token_id, end_token_id, max_length, 1.0, 0, 0.0)
        return self.beam_search_batch(encoder_output, start_token_id, end_token_id, max_length, beam_size)[0]

    def beam_search_batch(self, encoder_output: torch.Tensor, start_token_id: int, end_token_id: int,
                          max_length: int, beam_size: int, return_scores: bool = False, flags: int = 0):
        """N independent batch-1 beam searches in one launch (BASELINE config 3): element j
        equals ``inference(image[j:j+1], beam_size=k)`` of the reference.  ``flags``: _lib.FLAG_NO_GROUP selects
        the one-workgroup-per-image kernel (the automatic fallback when the grouped kernel times out)."""
        if beam_size > _lib.MAX_BEAM:
            raise NotImplementedError(f"img2latex_amd: beam_size <= {_lib.MAX_BEAM} (got {beam_size})")
        dec = self.decoder
        w, keep, enc = dec.prepare(encoder_output)
        n, dev = enc.shape[0], enc.device
        L = _lib.lib()
        nbytes = L.i2l_beam_workspace_bytes(n, beam_size, dec.hidden_dim, dec.lstm_layers, max_length)
        bws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        seq = torch.empty((n, max_length + 1), dtype=torch.int32, device=dev)
        ln = torch.empty((n,), dtype=torch.int32, device=dev)
        score = torch.empty((n,), dtype=torch.float64, device=dev)
        def launch(fl):
            _lib.check(L.i2l_beam_decode(ctypes.byref(w), dec._ws.data_ptr(), n, beam_size, max_length,
                                         int(start_token_id), int(end_token_id), bws.data_ptr(), nbytes,
                                         seq.data_ptr(), ln.data_ptr(), score.data_ptr(), int(fl), _lib.stream_ptr()),
                       "beam_decode")
        launch(flags | int(dec.kernel_flags))
        seq_h, ln_h = seq.cpu(), ln.cpu()
        if min(ln_h.tolist()) <= -3:
            # the grouped kernel needs its four workgroups resident together; on a GPU shared with other work a
            # poll can time out (len -3): run the one-workgroup-per-image kernel instead
            warnings.warn("img2latex_amd: grouped beam search timed out, re-running with one workgroup per image")
            launch(flags | int(dec.kernel_flags) | _lib.FLAG_NO_GROUP)
            seq_h, ln_h = seq.cpu(), ln.cpu()
        del keep
        lens = ln_h.tolist()                                   # one conversion each, then plain list slices
        out = [row[:ln_j] for row, ln_j in zip(seq_h.tolist(), lens)]
        if return_scores:
            return out, score.cpu().tolist()
        return out
