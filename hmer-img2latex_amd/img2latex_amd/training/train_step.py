"""One optimisation step of the reference's Trainer, fp32 branch (trainer.py:303-343):

    targets = formulas[:, 1:]                                        :306
    outputs = model(images, formulas)                                :334
    loss = CrossEntropyLoss(ignore_index=PAD, label_smoothing=0.1)   :111-115,335-336
    loss.backward(); clip_grad_norm_(5.0); Adam(lr, weight_decay).step(); zero_grad   :337-343

run entirely with the HIP kernels: training forward of encoder and decoder, fused CE, BPTT,
conv/linear backward writing straight into ONE flat gradient buffer, one all-reduce of that
buffer when data-parallel, then a fused clip + Adam over the flat parameter buffer.  No host
synchronisation inside the step; ``loss`` / ``total_norm`` stay on the device until read.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import _lib
from ..model._train_fn import (decoder_train_backward, decoder_train_forward, encoder_train_backward,
                               encoder_train_forward)
from .dp import OverlappedAllReduce, broadcast_parameters, rank_of


class TrainStep:
    def __init__(self, model, lr: float = 1e-3, weight_decay: float = 1e-4, clip_grad_norm: float = 5.0,
                 pad_token_id: int = 0, label_smoothing: float = 0.1, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, seed: int = 0, overlap_all_reduce: bool = True, side_wgrad: bool = True):
        self.model = model
        self.lr, self.weight_decay, self.clip = lr, weight_decay, clip_grad_norm
        self.pad, self.smoothing, self.betas, self.eps = pad_token_id, label_smoothing, betas, eps
        self.group = process_group
        self.step_count = 0
        self.seed = seed
        self.rank = rank_of(process_group)           # mixed into the dropout seed: every rank draws its own masks
        params = [p for p in model.parameters()]
        if not params or not params[0].is_cuda:
            raise RuntimeError("img2latex_amd: TrainStep needs the model on a ROCm device (no CPU fallback)")
        dev = params[0].device
        # 16-byte aligned slices of one flat buffer; parameters become views into it (state_dict unchanged).  The conv
        # stack's parameters go LAST: their gradients are the last the backward pass produces, so everything in front
        # of them can be all-reduced while the conv backward still runs (dp.OverlappedAllReduce)
        self.offsets: Dict[str, int] = {}
        # only what requires a gradient is optimised: torch.optim.Adam skips parameters whose .grad is None, i.e. the
        # frozen part of a ResNet backbone (encoder.py:201-210) gets neither an update nor weight decay
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        is_late = lambda n: n.startswith("encoder.cnn_layers.") or n.startswith("encoder.resnet.")
        late = [(n, p) for n, p in named if is_late(n)]
        early = [(n, p) for n, p in named if not is_late(n)]
        off = 0
        for name, p in early + late:
            if late and name == late[0][0]:
                self.n_early = off
            self.offsets[name] = off
            off += (p.numel() + 3) // 4 * 4
        self.n = off
        if not late:
            self.n_early = off
        self.flat_params = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(self.n + 4, dtype=torch.float32, device=dev)   # [n] loss sum, [n+1] count
        self.exp_avg = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.stats = torch.zeros(4, dtype=torch.float32, device=dev)
        self.grad_views: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for name, p in named:
                o = self.offsets[name]
                view = self.flat_params[o:o + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                self.grad_views[name] = self.flat_grads[o:o + p.numel()].view_as(p)
        broadcast_parameters(self.flat_params, 0, self.group)
        # overlap_all_reduce=False: literally ONE all-reduce of the whole flat buffer after the backward pass (BASELINE
        # north_star's wording); True (default): the same element-wise sums issued in two pieces, the first -- everything
        # but the conv gradients -- as soon as the FC backward is enqueued, beside ~1 ms of conv backward (dp.py)
        self.overlap_all_reduce = bool(overlap_all_reduce)
        # weight gradients on two side streams OWNED BY THIS OBJECT beside the data-gradient chain (csrc/api.hip: i2l_lanes;
        # r03 used streams hidden inside the library); False keeps the whole backward on one stream (same kernels, same
        # sums either way)
        self.side_wgrad = bool(side_wgrad)
        self._lanes = _lib.Lanes(dev, 2) if self.side_wgrad else None
        self._side_keep = []          # buffers the side lanes' kernels still read; released after the join
        self._accum = None            # gradient accumulation (micro_step): sum over micro-batches of grad(mean loss) / k
        self._micro = 0               # micro-batches accumulated since the last update
        self._ar_stream = None        # helper stream the early all-reduce piece is issued from (data parallel only)
        self._reducer = OverlappedAllReduce(self.flat_grads, self.n_early, self.group)
        L = _lib.lib()
        # zeroed once: besides scratch it carries the count of skipped (non-finite) updates across calls
        self._opt_ws = torch.zeros(L.i2l_optimizer_workspace_bytes(), dtype=torch.uint8, device=dev)

    # ------------------------------------------------------------------
    def forward_backward(self, images: torch.Tensor, formulas: torch.Tensor) -> torch.Tensor:
        """Training forward + backward for this rank's shard; fills flat_grads (gradient of the SUM of the
        per-token losses), flat_grads[n] = loss sum, flat_grads[n+1] = non-PAD count.  Returns logits."""
        model = self.model
        L = _lib.lib()
        formulas = formulas.to(torch.int32) if formulas.dtype != torch.int32 else formulas
        tokens_in = formulas[:, :-1].contiguous()                    # seq2seq.py:115-120
        targets = formulas[:, 1:].contiguous()                       # trainer.py:306
        B, T = tokens_in.shape
        # one dropout stream per (seed, optimizer step, data-parallel rank): the kernels hash the LOCAL element index,
        # so without the rank every replica would reuse rank 0's masks for its own rows
        self.step_seed = ((self.seed * 1000003 + self.step_count) * 4099 + self.rank + 7919 * self._micro) & 0x3FFFFFFFFFFFFFFF
        enc, enc_state = encoder_train_forward(model.encoder, images)
        logits, dec_state = decoder_train_forward(model.decoder, enc, tokens_in, self.step_seed)
        V = logits.shape[-1]
        dlogits = torch.empty_like(logits)
        nbytes = L.i2l_ce_workspace_bytes(B * T)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=logits.device)
        _lib.check(L.i2l_ce_label_smooth_fwd_bwd(logits.data_ptr(), targets.data_ptr(), B * T, V, self.pad,
                                                 self.smoothing, ws.data_ptr(), nbytes, dlogits.data_ptr(),
                                                 self.flat_grads.data_ptr() + 4 * self.n, _lib.stream_ptr()),
                   "ce_label_smooth_fwd_bwd")
        dgr = {n[len("decoder."):]: g for n, g in self.grad_views.items() if n.startswith("decoder.")}
        # weight gradients feed nothing in the backward chain: they go to this object's side lanes and fill the launch
        # gaps and tile tails of the data-gradient chain; the join at the end of this method orders them before the
        # all-reduce / Adam.  Everything those kernels read is held in _side_keep until then.
        side = self._lanes.handle if self.side_wgrad else None
        self._join_side()                                            # a previous forward_backward() without apply()
        denc = decoder_train_backward(model.decoder, dec_state, dlogits, dgr, lanes=side)
        if side:
            self._side_keep.extend((dec_state, dlogits, enc_state, images, tokens_in))
        egr = {n[len("encoder."):]: g for n, g in self.grad_views.items() if n.startswith("encoder.")}

        def after_linear():
            # decoder + FC gradients are final once the FC backward is enqueued -- and, with side-stream weight
            # gradients, once the side lanes have produced them: the early all-reduce is issued from a helper stream that
            # waits for both, so that the conv backward on the step's own stream is not held up by the join
            if not self._reducer._active():
                return
            if not self._side_keep:
                self._reducer.start_early()
                return
            main = torch.cuda.current_stream()
            if self._ar_stream is None:
                self._ar_stream = torch.cuda.Stream(device=self.flat_grads.device)
            self._ar_stream.wait_stream(main)
            with torch.cuda.stream(self._ar_stream):
                self._lanes.join()
                self._reducer.start_early()
        encoder_train_backward(model.encoder, enc_state, denc, egr,
                               after_linear=after_linear if self.overlap_all_reduce else None, lanes=side,
                               keep=self._side_keep)
        self._join_side()          # flat_grads is complete for whatever the current stream does next (all-reduce, Adam, a reader)
        return logits

    def _join_side(self, clear: bool = True) -> None:
        """The current stream waits for the side lanes' weight gradients (no-op when none are pending)."""
        if self._side_keep:
            self._lanes.join()
            if clear:
                # the buffers may be reused by later work on this stream only: that work is ordered after the join
                self._side_keep = []

    def apply(self) -> None:
        """All-reduce (data parallel), then clip + Adam; bumps the step counter."""
        L = _lib.lib()
        self._join_side()
        self._reducer.finish()
        self.step_count += 1
        _lib.check(L.i2l_grad_clip_adam_step(
            self.flat_params.data_ptr(), self.flat_grads.data_ptr(), self.exp_avg.data_ptr(),
            self.exp_avg_sq.data_ptr(), self.n, self.flat_grads.data_ptr() + 4 * (self.n + 1), float(self.clip),
            float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay),
            self.step_count, self._opt_ws.data_ptr(), self._opt_ws.numel(), self.stats.data_ptr(),
            _lib.stream_ptr()), "grad_clip_adam_step")
        # the kernel wrote the parameters through raw pointers: tell autograd / the weight-image caches
        # (LSTMDecoder.prepare, ResNetEncoder) that they changed
        for p in self.model.parameters():
            torch.autograd.graph.increment_version(p)

    def step(self, images: torch.Tensor, formulas: torch.Tensor) -> Dict[str, torch.Tensor]:
        """trainer.py:303-343 for one batch.  Returns device tensors (no sync): loss, total_norm, count, skipped.
        ``skipped`` is 1.0 when the gradients were not finite (e.g. a grouped recurrence kernel timed out on a shared
        GPU and poisoned its outputs with NaN): the fused clip + Adam kernel then leaves parameters and moments
        untouched -- on every rank, since the all-reduce spreads the NaN -- and the caller may simply repeat the
        batch, if it keeps happening with ``model.decoder.kernel_flags |= FLAG_NO_GROUP``."""
        self.model.train()
        self.forward_backward(images, formulas)
        self.apply()
        count = self.flat_grads[self.n + 1]
        return dict(loss=self.flat_grads[self.n] / count.clamp(min=1.0), total_norm=self.stats[0], count=count,
                    skipped=self.stats[3])

    def micro_step(self, images: torch.Tensor, formulas: torch.Tensor, accumulation_steps: int, update: bool) -> Dict[str, torch.Tensor]:
        """Gradient accumulation, trainer.py:345-383 (the shipped configuration: accumulation_steps 4): every micro-batch
        contributes the gradient of ITS mean loss divided by ``accumulation_steps`` (``loss = criterion(...) /
        accumulation_steps; loss.backward()``), and with ``update`` -- every accumulation_steps-th batch or the last one of
        the epoch, :374 -- the sum is clipped and Adam steps.  Data parallel: a micro-batch's mean is over the GLOBAL
        micro-batch, so its [loss sum, count] pair is all-reduced at once (8 bytes) and the accumulated gradients once, at
        the update.  Returns the micro-batch's mean loss (what the reference logs, :386-388) and, after an update,
        ``total_norm`` / ``skipped``."""
        import torch.distributed as dist
        if accumulation_steps < 1:
            raise ValueError("accumulation_steps must be >= 1")
        if accumulation_steps == 1 and self._micro == 0 and update:
            return self.step(images, formulas)
        self.model.train()
        overlap, self.overlap_all_reduce = self.overlap_all_reduce, False      # nothing of a micro-batch is reduced early
        try:
            self.forward_backward(images, formulas)
        finally:
            self.overlap_all_reduce = overlap
        tail = self.flat_grads[self.n:self.n + 2].clone()                      # [loss sum, count] of this micro-batch
        if self._reducer._active():
            dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
        inv = 1.0 / (tail[1].clamp(min=1.0) * float(accumulation_steps))
        if self._accum is None:
            self._accum = torch.zeros(self.n, dtype=torch.float32, device=self.flat_grads.device)
        self._accum.addcmul_(self.flat_grads[:self.n], inv.expand(self.n))
        self._micro += 1
        out = dict(loss=tail[0] / tail[1].clamp(min=1.0), count=tail[1])
        if update:
            world = dist.get_world_size(self.group) if self._reducer._active() else 1
            self.flat_grads[:self.n].copy_(self._accum)
            self.flat_grads[self.n] = out["loss"] / world                      # the slots the fused kernel divides by: the sum
            self.flat_grads[self.n + 1] = 1.0 / world                          # over ranks of the count slot is exactly 1
            self._accum.zero_()
            self._micro = 0
            self.apply()
            out.update(total_norm=self.stats[0], skipped=self.stats[3])
        return out

    def applied_steps(self) -> int:
        """Optimizer steps actually applied (host sync): step_count minus the skipped, non-finite ones.  Called at
        checkpoint / logging time; warns when every step since the last call was skipped (gradients that STAY
        non-finite are divergence, not a one-off timeout -- the silent no-op updates would otherwise go unnoticed)."""
        skipped = int(self._opt_ws[-256:-252].view(torch.int32).item())
        last_steps, last_skipped = getattr(self, "_last_counts", (0, 0))
        if self.step_count - last_steps >= 8 and skipped - last_skipped == self.step_count - last_steps:
            import warnings
            warnings.warn(f"img2latex_amd: all {self.step_count - last_steps} optimizer steps since the last check were "
                          "skipped (non-finite gradients): the run has diverged or the grouped kernels keep timing out")
        self._last_counts = (self.step_count, skipped)
        return self.step_count - skipped

    # ------------------------------------------------------------------ checkpoint compatibility
    def optimizer_state_dict(self) -> Dict:
        """torch.optim.Adam.state_dict() layout (what trainer.py:213 stores): resuming in the reference works."""
        state, params = {}, []
        applied = self.applied_steps()
        for i, (name, p) in enumerate(self.model.named_parameters()):
            if name not in self.offsets:                      # frozen: torch's Adam holds no state for it either
                params.append(i)
                continue
            o = self.offsets[name]
            state[i] = {"step": torch.tensor(float(applied)),
                        "exp_avg": self.exp_avg[o:o + p.numel()].view_as(p).detach().cpu().clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view_as(p).detach().cpu().clone()}
            params.append(i)
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False, "params": params}
        # like torch's Adam under GradScaler, the state stays empty until an update has actually been APPLIED
        return {"state": state if applied > 0 else {}, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: Dict) -> None:
        """Inverse of the above (trainer.py:255, resume from a reference checkpoint)."""
        names = [n for n, _ in self.model.named_parameters()]
        for i, st in sd.get("state", {}).items():
            name = names[int(i)]
            if name not in self.offsets:
                continue
            p = dict(self.model.named_parameters())[name]
            o = self.offsets[name]
            self.exp_avg[o:o + p.numel()].view_as(p).copy_(st["exp_avg"])
            self.exp_avg_sq[o:o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
            self.step_count = int(float(st["step"]))
            self._opt_ws.zero_()
        g = sd.get("param_groups", [{}])[0]
        self.lr = g.get("lr", self.lr)
        self.weight_decay = g.get("weight_decay", self.weight_decay)
