"""Two-stage batch pipeline for greedy inference on one GPU.

Images are independent, so consecutive batches can overlap: the encoder of batch i + 1 runs on one HIP stream while the
persistent decode loop of batch i runs on another.  Two ways of sharing the chip are built:

* co-resident (r03, what bench.py times and Predictor uses): ``decode_flags = FLAG_DECODE_GROUP16`` (r04: 16 members x 16
  rows per group, the per-step products as split-bf16 MFMA bursts that resident conv waves barely stretch) or
  ``FLAG_DECODE_GROUP8`` (r03: 8 x 8 on the vector ALUs) -- either keeps one wave per SIMD and 50 - 80 KB of LDS on EVERY
  compute unit, which leaves room for one conv workgroup of the next batch's encoder on the same unit: the two kernels share
  the chip by resource.  The encoder of batch i + 1 is held
  back until the decode of batch i owns its compute units -- a DEPENDENCY since r04: the decode kernel publishes a
  sequence number once all its groups are resident (i2l_greedy_decode_ex's residency signal) and the encoder stream
  waits for that word (i2l_stream_wait_value32); r03 guessed the moment with a 30 us delay kernel;
* by compute-unit count (r01, ``rows_per_workgroup = 2``): a row-per-workgroup decode on 128 units, the encoder on the
  other 128.  2.6x slower than the grouped kernels; kept for decoders the grouped kernels do not cover.

Nothing is skipped: every batch still runs the full encoder, decoder prepare, decode loop
and id copy; results equal ``Seq2SeqModel._greedy_search`` batch by batch.
"""
from __future__ import annotations

from collections import deque
from typing import Deque, List, Optional, Tuple

import torch

from . import _lib


class GreedyPipeline:
    """submit(images) enqueues one batch; results come back in order from collect()."""

    def __init__(self, model, start_token_id: int, end_token_id: int, max_length: int = 150,
                 temperature: float = 1.0, depth: int = 3, rows_per_workgroup: int = 2, decode_streams: int = 1,
                 decode_flags: int = 0, encoder_flags: int = 0, decode_priority: int = 0, encoder_streams: int = 1,
                 encoder_priority: int = 0, hold_encoder: Optional[bool] = None, wait_timeout_us: float = 20000.0,
                 clear_early: bool = True,
                 stop: int = _lib.STOP_NONE, select: int = _lib.SELECT_LOGITS):
        self.model = model
        self.start, self.end, self.max_length, self.temperature = start_token_id, end_token_id, max_length, temperature
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("img2latex_amd: GreedyPipeline needs the model on a ROCm device (no CPU fallback)")
        self.device = dev
        # several encoder streams (round robin): the encoders of consecutive batches run side by side, each one's launch
        # gaps and tile tails filled by the other's workgroups (pays for the 53-launch ResNet trunk)
        self.enc_streams = [torch.cuda.Stream(device=dev, priority=int(encoder_priority)) for _ in range(max(1, encoder_streams))]
        self._next_enc = 0
        # several decode streams (round robin) let decodes of consecutive batches run side by side, each on
        # few CUs (more rows per workgroup = less weight traffic per row); depth must cover them
        self.dec_streams = [torch.cuda.Stream(device=dev, priority=int(decode_priority)) for _ in range(max(1, decode_streams))]
        self._next_dec = 0
        # depth 3 (default since r04): with two batches in flight the host can enqueue encoder(i + 1) only after it has collected
        # batch i - 1, ~100 us into decode(i) (a kernel trace showed the encoder queue idle that long); with three its kernels
        # are already queued behind the residency wait when decode(i) is launched
        self.depth = max(depth, len(self.dec_streams) + len(self.enc_streams))
        self.rows_per_workgroup = rows_per_workgroup     # 2: decode occupies half of the CUs, the encoder the rest
        # r03: decode_flags = _lib.FLAG_DECODE_GROUP8 (with rows_per_workgroup = 0) runs the 8-member grouped decode
        # instead -- one wave per SIMD and 80 KB of LDS on EVERY CU, which leaves room for one conv workgroup of the next
        # batch's encoder on the same CU: the two kernels share the chip by resource, not by CU count
        self.decode_flags = int(decode_flags)
        self.stop, self.select = int(stop), int(select)   # STOP_STICKY + SELECT_SOFTMAX = Predictor.predict_batch's loop
        self.encoder_flags = int(encoder_flags)          # e.g. _lib.FLAG_CONV_ONE_PER_CU while the pipeline runs
        if len(self.enc_streams) >= 2:
            # several trunks in flight: a trunk's idle CUs are filled by its neighbour, so the shapes that balance ONE layer over
            # the chip (64-column tiles, the LDS-patch 3x3 kernel: -4 % on a trunk alone) only cost arithmetic intensity here --
            # same box, two trunks + decode: 16.7 M tokens/s with them, 17.0 M with 128-column ring tiles (profiles/r04/resnet_patch.txt)
            # (ResNetEncoder.MULTI_STREAM_FLAGS = FLAG_RESNET_WIDE_TILES | FLAG_RESNET_NO_PATCH; the CNN encoder has none)
            self.encoder_flags |= getattr(model.encoder, "MULTI_STREAM_FLAGS", 0)
        # Co-resident mode: the encoder of batch i + 1 is held back until the decode of batch i is RESIDENT, so that the
        # decode's 256 workgroups are on the compute units before the first conv workgroup asks for one.  Launched the other
        # way round -- conv0 of batch i + 1 beside prepare(i), conv1 racing the decode for CUs -- the two kernels fall into a
        # schedule in which conv1 crawls for the whole decode and conv2 / FC follow alone: 1.25 - 1.45 ms per batch instead of
        # 1.07 (profiles/r03/ramp.txt).  The decode launch of batch j publishes j + 1 to `_resident` when its last group has
        # passed the placement exchange; with n encoder streams the encoder of batch j waits for the value of decode(j - n):
        # every encoder starts at a decode's residency, n of them in flight.  `hold_encoder=False` switches the wait off (A/B);
        # `wait_timeout_us` bounds it (a launch that fell back to an ungrouped kernel publishes at once, a timed-out group
        # never does).
        if hold_encoder is None:
            hold_encoder = bool(self.decode_flags & (_lib.FLAG_DECODE_GROUP8 | _lib.FLAG_DECODE_GROUP16))
        self.hold_encoder = bool(hold_encoder)
        self.clear_early = bool(clear_early)             # the decode's group region is zeroed on the encoder stream (see submit)
        self.wait_timeout_us = float(wait_timeout_us)
        self._resident = torch.zeros(1, dtype=torch.int32, device=dev)
        # measurement hook (bench.py): a list here receives, per submitted batch, the timing events {"enc_start" (behind the
        # wait), "enc_end" (behind prepare), "dec_start", "dec_end" (around the decode launch on ITS stream)}
        self.trace: Optional[list] = None
        self.trace_every = 1         # with `trace`: batches j and j + 1 for every j that is a multiple of this (1 = every batch); a timing
                                     # event is a marker packet in the queue, and four per batch cost the pipeline ~1 % (r04)
        self._inflight: Deque[Tuple[torch.cuda.Event, torch.Tensor, torch.Tensor]] = deque()
        self.copy_stream = torch.cuda.Stream(device=dev)     # the ids' device -> host copies (off the decode stream: +1 %)
        self._batch_no = 0
        self._slots = self.depth + 1                 # decoder workspaces in rotation: a slot is rewritten only after its batch was collected
        self._free: List[torch.Tensor] = []          # pinned host buffers not in use
        self._lent: Optional[torch.Tensor] = None    # buffer handed to the caller by the last collect()

    def submit(self, images: torch.Tensor) -> None:
        """Enqueue encoder + decoder prepare (stream A) and decode + id copy (stream B) for one batch."""
        if len(self._inflight) >= self.depth:
            raise RuntimeError(f"GreedyPipeline: {self.depth} batches already in flight; collect() first")
        cur = torch.cuda.current_stream(self.device)
        enc_stream = self.enc_streams[self._next_enc]
        self._next_enc = (self._next_enc + 1) % len(self.enc_streams)
        enc_stream.wait_stream(cur)                           # images were produced on the caller's stream
        with torch.no_grad():
            with torch.cuda.stream(enc_stream):
                if self.hold_encoder and self._batch_no >= len(self.enc_streams):
                    # decode(j - n) is resident <=> the word has reached (j - n) + 1
                    _lib.check(_lib.lib().i2l_stream_wait_value32(
                        self._resident.data_ptr(), self._batch_no - len(self.enc_streams) + 1, self.wait_timeout_us,
                        _lib.stream_ptr()), "stream_wait_value32")
                rec = None
                if self.trace is not None and self._batch_no % self.trace_every < 2:      # pairs of consecutive batches
                    rec = {k: torch.cuda.Event(enable_timing=True) for k in ("enc_start", "enc_end", "dec_start", "dec_end")}
                    rec["batch"] = self._batch_no
                    rec["enc_start"].record(enc_stream)
                    self.trace.append(rec)
                saved = self.model.encoder.kernel_flags
                self.model.encoder.kernel_flags = saved | self.encoder_flags
                try:
                    enc = self.model.encoder(images)
                finally:
                    self.model.encoder.kernel_flags = saved
                # the decoder's per-batch tables (i2l_decoder_prepare) are built here, on the ENCODER stream right behind the FC
                # layer, into one of `depth + 1` workspaces: the decode stream then goes from decode(i) straight into
                # decode(i + 1) (the two small prepare launches used to sit between them, ~50 us per batch)
                slot = self._batch_no % self._slots
                seq = self._batch_no + 1
                self._batch_no += 1
                dec = self.model.decoder
                w, keep, enc_c = dec.prepare(enc, slot=("pipe", id(self), slot))
                prepared = (w, keep, enc_c, dec._ws)
                # the grouped launch zeroes its status words + exchange granules first; done HERE, behind prepare on the encoder
                # stream (which ends before the running decode does), that memset is no longer between two decodes on their queue
                dflags = self.decode_flags
                if self.clear_early and dflags & (_lib.FLAG_DECODE_GROUP8 | _lib.FLAG_DECODE_GROUP16):
                    dims = (enc.shape[0], dec.vocab_size, dec.embedding_dim, dec.hidden_dim, dec.lstm_layers)
                    off = _lib.lib().i2l_decoder_group_status_offset(*dims)
                    nb = _lib.lib().i2l_decoder_group_region_bytes(*dims)
                    if off and nb:
                        dec._ws[off:off + nb].zero_()
                        dflags |= _lib.FLAG_DECODE_REGION_CLEARED
                enc_done = torch.cuda.Event() if rec is None else rec["enc_end"]
                enc_done.record(enc_stream)
            images.record_stream(enc_stream)
            dec_stream = self.dec_streams[self._next_dec]
            self._next_dec = (self._next_dec + 1) % len(self.dec_streams)
            with torch.cuda.stream(dec_stream):
                dec_stream.wait_event(enc_done)
                enc.record_stream(dec_stream)
                if rec is not None:
                    rec["dec_start"].record(dec_stream)
                ids, _ = self.model.greedy_ids(enc, self.start, self.end, self.max_length, self.temperature,
                                               stop=self.stop, select=self.select,
                                               rows_per_workgroup=self.rows_per_workgroup, flags=dflags,
                                               prepared=prepared,
                                               resident=(self._resident, seq) if self.hold_encoder else None)
                if rec is not None:
                    rec["dec_end"].record(dec_stream)
                host = self._host_buffer(ids.shape)
                if self.copy_stream is None:
                    host.copy_(ids, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(dec_stream)
                else:                                              # the id copy leaves the decode stream: decode(i + 1) need not wait for it
                    decoded = torch.cuda.Event()
                    decoded.record(dec_stream)
                    self.copy_stream.wait_event(decoded)
                    with torch.cuda.stream(self.copy_stream):
                        host.copy_(ids, non_blocking=True)
                        done = torch.cuda.Event()
                        done.record(self.copy_stream)
                    ids.record_stream(self.copy_stream)
        self._inflight.append((done, host, enc))

    def close(self) -> None:
        """Give the decoder workspaces of this pipeline's slots back (they live in the decoder's cache under this
        pipeline's id); called by Predictor.predict_ids_stream when its stream ends, and on garbage collection."""
        dec = getattr(getattr(self, "model", None), "decoder", None)
        if dec is not None and hasattr(dec, "release_slots"):
            dec.release_slots(id(self))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _host_buffer(self, shape) -> torch.Tensor:
        for i, t in enumerate(self._free):
            if t.shape == shape:
                return self._free.pop(i)
        return torch.empty(shape, dtype=torch.int32).pin_memory()

    def pending(self) -> int:
        return len(self._inflight)

    def collect(self) -> Optional[torch.Tensor]:
        """Oldest finished batch: (B, steps) int32 ids in pinned host memory, valid until the next
        collect(); ``to_sequences`` applies the reference's stop rule and list conversion."""
        if not self._inflight:
            return None
        done, host, enc = self._inflight.popleft()
        done.synchronize()
        if _lib.ids_timed_out(host):
            # a grouped decode needs its members resident together; on a GPU shared with other work a bounded wait can
            # expire (ids -3): this batch is decoded again on the row-per-workgroup kernel, which needs no partner
            import warnings
            warnings.warn("img2latex_amd: grouped decode timed out inside the pipeline; re-running the batch on the "
                          "row-per-workgroup kernel", RuntimeWarning)
            with torch.no_grad():
                ids, _ = self.model.greedy_ids(enc, self.start, self.end, self.max_length, self.temperature,
                                               stop=self.stop, select=self.select, rows_per_workgroup=1)
            host.copy_(ids)
        if self._lent is not None:
            self._free.append(self._lent)
        self._lent = host
        return _lib.check_ids(host)

    def drain(self) -> List[torch.Tensor]:
        out = []
        while self._inflight:
            out.append(self.collect().clone())
        return out

    def to_sequences(self, ids_host: torch.Tensor):
        """The reference's post-processing of ``_greedy_search`` (seq2seq.py:217-232) for B > 1."""
        all_end = (ids_host == self.end).all(dim=0)
        steps = int(all_end.nonzero()[0]) + 1 if bool(all_end.any()) else ids_host.shape[1]
        return [[self.start] + r for r in ids_host[:, :steps].tolist()]
