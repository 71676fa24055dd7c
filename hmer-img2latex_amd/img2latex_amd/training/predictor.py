"""Predictor with the reference's surface (img2latex/training/predictor.py:20-394) on the HIP path.

Checkpoint compatibility: ``from_checkpoint`` reads the dict the reference's
``Trainer.save_checkpoint`` writes (trainer.py:209-224: ``model_state_dict``, ``config``,
``tokenizer_config{token_to_id, special_tokens, max_sequence_length}``) and rebuilds model and
token table exactly as predictor.py:80-129 does.  The greedy loop of ``predict_batch``
(predictor.py:283-358) runs as ONE persistent kernel launch (I2L_STOP_STICKY + argmax of softmax).

``predict`` / ``predict_batch`` take what the reference's take -- a path, a PIL image, a numpy array or a tensor
-- through ``_prepare_image`` (predictor.py:396-462, incl. its hard-coded 64x800 target and its range-dependent
normalisation of tensors); the pixels are resampled on the device (img2latex_amd.data).  ``predict_batch_ids`` /
``evaluate_batch`` are this package's batch entry points for tensors ALREADY shaped for the model (the benchmark
drivers: SURVEY 8b "feed tensors straight to the encoder").
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from .. import _lib
from ..model import Seq2SeqModel

DEFAULT_SPECIAL_TOKENS = {"PAD": "<PAD>", "START": "<START>", "END": "<END>", "UNK": "<UNK>"}   # tokenizer.py:36-41


class TokenTable:
    """The part of LaTeXTokenizer the predictor touches: id maps, special ids, decode()
    (tokenizer.py:68-78,166-192).  String <-> id only; fitting a vocabulary is out of scope."""

    def __init__(self, token_to_id: Optional[Dict[str, int]] = None, special_tokens: Optional[Dict[str, str]] = None,
                 max_sequence_length: int = 141):
        self.special_tokens = dict(special_tokens or DEFAULT_SPECIAL_TOKENS)
        if token_to_id is None:
            token_to_id = {tok: i for i, tok in enumerate(self.special_tokens.values())}
        self.token_to_id = dict(token_to_id)
        self.id_to_token = {i: t for t, i in self.token_to_id.items()}
        self.vocab_size = len(self.token_to_id)
        self.max_sequence_length = max_sequence_length
        self.pad_token_id = self.token_to_id[self.special_tokens["PAD"]]
        self.start_token_id = self.token_to_id[self.special_tokens["START"]]
        self.end_token_id = self.token_to_id[self.special_tokens["END"]]
        self.unk_token_id = self.token_to_id[self.special_tokens["UNK"]]

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        special = {self.token_to_id[t] for t in self.special_tokens.values()} if skip_special_tokens else set()
        return " ".join(self.id_to_token.get(int(i), self.special_tokens["UNK"]) for i in ids if int(i) not in special)

    def config(self) -> Dict:
        return {"token_to_id": self.token_to_id, "special_tokens": self.special_tokens,
                "max_sequence_length": self.max_sequence_length}


def model_from_checkpoint_config(config: Dict, vocab_size: int) -> Seq2SeqModel:
    """predictor.py:83-126: config["model"] -> Seq2SeqModel constructor arguments."""
    model_config = config.get("model", {})
    model_type = model_config.get("name", "cnn_lstm")
    enc = dict(model_config.get("encoder", {}).get("cnn" if model_type == "cnn_lstm" else "resnet", {}))
    enc["embedding_dim"] = model_config.get("embedding_dim", 256)
    return Seq2SeqModel(model_type=model_type, vocab_size=vocab_size, encoder_params=enc,
                        decoder_params=model_config.get("decoder", {}))


def save_checkpoint(path: str, model: Seq2SeqModel, tokenizer: TokenTable, config: Dict, epoch: int = 0, step: int = 0,
                    metrics: Optional[Dict] = None, optimizer_state_dict: Optional[Dict] = None) -> None:
    """The dict of trainer.py:209-224, readable by the reference's Predictor.from_checkpoint / load_checkpoint."""
    torch.save({"epoch": epoch, "step": step,
                "model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                "optimizer_state_dict": optimizer_state_dict or {}, "metrics": metrics or {}, "config": config,
                "tokenizer_config": tokenizer.config()}, path)


def _is_pil(obj) -> bool:
    try:
        from PIL import Image
    except ImportError:                                                     # pragma: no cover
        return False
    return isinstance(obj, Image.Image)


class Predictor:
    def __init__(self, model: Seq2SeqModel, tokenizer, device: Optional[torch.device] = None,
                 model_type: str = "cnn_lstm"):
        self.device = torch.device("cuda") if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("img2latex_amd: Predictor runs on a ROCm device only (no CPU fallback)")
        self.model = model.to(self.device)
        self.tokenizer = tokenizer
        self.model_type = model_type
        self.model.eval()                                                   # predictor.py:55

    @classmethod
    def from_checkpoint(cls, checkpoint_path: str, device: Optional[torch.device] = None) -> "Predictor":
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        config = checkpoint.get("config", {})
        tcfg = checkpoint.get("tokenizer_config", {})
        tokenizer = TokenTable(tcfg.get("token_to_id"), tcfg.get("special_tokens"), tcfg.get("max_sequence_length", 141))
        model = model_from_checkpoint_config(config, tokenizer.vocab_size)
        model.load_state_dict(checkpoint["model_state_dict"])               # predictor.py:129, strict
        return cls(model=model, tokenizer=tokenizer, device=device,
                   model_type=config.get("model", {}).get("name", "cnn_lstm"))

    # ------------------------------------------------------------------ predictor.py:396-521
    def _prepare_image(self, image) -> torch.Tensor:
        """One image of any accepted type -> (1, C, 64, 800) on the device, with the reference's rules per type:
        str -> load_image (aspect-preserving LANCZOS, pad / crop, [-1,1] or ImageNet statistics; a missing file gives
        a zero image); tensor / ndarray -> bilinear resize when the size differs, ``/255*2-1`` iff min < 0 or max > 1
        (:493-497 -- so a tensor already in [-1,1] is rescaled again, as in the reference); PIL image -> mode
        conversion, Image.resize((800, 64)) with Pillow's default BICUBIC (aspect NOT kept), ``/255*2-1``."""
        from ..data import load_image, preprocess_batch
        img_size = (64, 800)                                                # :409-414, both model types
        channels = 1 if self.model_type == "cnn_lstm" else 3
        if isinstance(image, str):
            t = load_image(image, img_size, channels).to(self.device)
        elif isinstance(image, torch.Tensor):
            t = self._preprocess_tensor(image, img_size, channels)
        elif isinstance(image, np.ndarray):
            t = self._preprocess_tensor(self._numpy_to_tensor(image, channels), img_size, channels)
        elif _is_pil(image):
            want = "L" if channels == 1 else "RGB"
            if image.mode not in ("L", "RGB"):                              # decoding-side conversion stays with PIL
                image = image.convert(want)
            with torch.cuda.device(self.device):
                t = preprocess_batch([np.array(image)], img_size, channels, "symmetric", keep_aspect=False,
                                     resample="bicubic")[0]
        else:
            raise TypeError(f"Unsupported image type: {type(image)}. "
                            "Expected str, torch.Tensor, numpy.ndarray, or PIL.Image.Image.")
        if self.model_type == "resnet_lstm" and t.shape[0] == 1:            # :453-455
            t = t.repeat(3, 1, 1)
        if t.dim() == 3:
            t = t.unsqueeze(0)
        return t

    def _preprocess_tensor(self, tensor: torch.Tensor, img_size, channels: int) -> torch.Tensor:
        """predictor.py:464-499 on the device."""
        from ..data import resize_bilinear
        tensor = tensor.to(self.device, dtype=torch.float32)
        if tensor.dim() == 2:
            tensor = tensor.unsqueeze(0)
        if tuple(tensor.shape[-2:]) != tuple(img_size):
            tensor = resize_bilinear(tensor.unsqueeze(0) if tensor.dim() == 3 else tensor, img_size)
            if tensor.dim() == 4 and tensor.shape[0] == 1:
                tensor = tensor.squeeze(0)
        if bool(tensor.min() < 0) or bool(tensor.max() > 1):
            tensor = tensor / 255.0
            tensor = tensor * 2.0 - 1.0
        return tensor

    @staticmethod
    def _numpy_to_tensor(array: np.ndarray, channels: int) -> torch.Tensor:
        """predictor.py:501-521: (H,W) -> (1,H,W); (H,W,C) -> (C,H,W) unless the first axis already is 1 or 3."""
        if array.ndim == 2:
            array = np.expand_dims(array, axis=0)
        elif array.ndim == 3 and array.shape[0] not in [1, 3]:
            array = np.transpose(array, (2, 0, 1))
        return torch.from_numpy(np.ascontiguousarray(array)).float()

    def _as_batch(self, images) -> torch.Tensor:
        if isinstance(images, torch.Tensor):
            t = images if images.dim() == 4 else images.unsqueeze(0)
        else:
            items = []
            for im in images:
                if not isinstance(im, torch.Tensor):
                    raise TypeError("img2latex_amd Predictor takes image TENSORS already sized for the model; file / "
                                    "PIL / numpy preprocessing (predictor.py:396-499) is host-side work outside this path")
                items.append(im if im.dim() == 3 else im.squeeze(0))
            t = torch.stack(items)
        return t.to(self.device, dtype=torch.float32).contiguous()

    def predict_batch_ids(self, images, max_length: int = 141, temperature: float = 1.0, top_k: int = 0,
                          top_p: float = 0.0, seed: Optional[int] = None) -> List[List[int]]:
        """Token ids of the batched loop (predictor.py:254-361): START kept, cut before the first END.
        Sampling (multinomial over the top-k / top-p masked softmax) iff temperature > 0 and
        (top_k > 0 or top_p > 0), exactly the reference's condition (:330); argmax of softmax otherwise."""
        x = self._as_batch(images)
        start, end = self.tokenizer.start_token_id, self.tokenizer.end_token_id
        with torch.no_grad():
            enc = self.model.encoder(x)
            if temperature > 0 and (top_k > 0 or top_p > 0.0):
                if seed is None:
                    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                tok0 = torch.full((enc.shape[0],), int(start), dtype=torch.int32, device=enc.device)
                ids, _ = self.model.decoder.sample_steps(enc, max_length, tok0, temperature, top_k, top_p, seed,
                                                         stop=_lib.STOP_STICKY, end_id=end)
                ids = _lib.check_ids(ids.cpu())
            else:       # same timeout fallback as Seq2SeqModel._greedy_search
                ids = self.model.greedy_ids_host(enc, start, end, max_length, temperature, stop=_lib.STOP_STICKY,
                                                 select=_lib.SELECT_SOFTMAX)
        a = ids.numpy()
        stop = (a == end) | (a < 0)                           # first END, or the kernel's "row finished" filler
        lens = np.where(stop.any(axis=1), stop.argmax(axis=1), a.shape[1]).tolist()
        return [[start] + row[:n] for row, n in zip(a.tolist(), lens)]

    def predict_ids_stream(self, batches, max_length: int = 141, temperature: float = 1.0):
        """``predict_batch_ids`` (the greedy branch: argmax of softmax, sticky stop) for a STREAM of image batches, through
        ``GreedyPipeline``: the decode of batch i shares the GPU with the encoder of batch i + 1 (two batches in flight, the
        schedule bench.py times).  Yields one ``List[List[int]]`` per batch, in order, equal to what ``predict_batch_ids``
        returns for that batch (up to fp32 near-ties between the two decode kernels)."""
        from ..pipeline import GreedyPipeline
        start, end = self.tokenizer.start_token_id, self.tokenizer.end_token_id
        n_enc = 2 if hasattr(self.model.encoder, "_trunk_train") else 1      # ResNet trunk: two in flight
        pipe = GreedyPipeline(self.model, start, end, max_length, temperature, rows_per_workgroup=0,
                              decode_flags=_lib.FLAG_DECODE_GROUP16, decode_priority=-1, encoder_streams=n_enc,
                              stop=_lib.STOP_STICKY, select=_lib.SELECT_SOFTMAX)

        def finish(ids_host):
            a = ids_host.numpy()
            stop = (a == end) | (a < 0)
            lens = np.where(stop.any(axis=1), stop.argmax(axis=1), a.shape[1]).tolist()
            return [[start] + row[:n] for row, n in zip(a.tolist(), lens)]

        try:
            for images in batches:
                if pipe.pending() >= pipe.depth:
                    yield finish(pipe.collect())
                pipe.submit(self._as_batch(images))
            while pipe.pending():
                yield finish(pipe.collect())
        finally:
            pipe.close()                                     # gives its decoder workspaces back

    def evaluate_batch(self, images, targets: torch.Tensor, max_length: Optional[int] = None) -> Dict:
        """One batch of the reference's ``evaluate`` command (cli.py:449-495) with every stage on the device and
        the token ids never leaving HBM between them:

            images    a (B,C,H,W) tensor already shaped for the model, or a list of decoded uint8 arrays
                      ((h,w) / (h,w,3)) that first go through ``preprocess_batch`` (= load_image, data/utils.py:18-90)
            encoder -> sticky greedy loop (predictor.py:283-358) -> i2l_compact_ids (cut before END, drop the special
            tokens: what tokenizer.decode + tokenizer.encode do to a prediction at cli.py:466-481; targets: drop PAD)
            -> i2l_sequence_metrics -> ONE small device->host copy of the integer statistics -> the reference's
            float64 formulas (calculate_metrics, metrics.py:184-223).

        Returns {"bleu", "levenshtein", "batch_size"} as ``calculate_metrics`` does, plus "pred_ids"/"pred_len"
        (device tensors) for callers that also want the strings."""
        return self._evaluate_finish(self._evaluate_launch(images, targets, max_length))

    def evaluate_stream(self, batches, max_length: Optional[int] = None, lookahead: int = 2):
        """``evaluate_batch`` over an iterable of (images, targets), one result per batch in order, software-pipelined:
        the host work of batch i+1 (shape pass, packing the ragged pages into pinned memory, their upload on a side
        stream) runs while the device is still busy with batch i, and the host only waits for batch i's statistics
        after batch i+1 has been enqueued.  On the device the two batches in flight share the GPU the way GreedyPipeline's
        do: preprocessing + encoder + decoder prepare of batch i+1 on one stream, the decode (8-member grouped kernel) +
        id compaction + statistics of batch i on another, the first held back until the second's decode is resident.
        ``lookahead`` batches are enqueued beyond the one whose statistics the host waits for (r04: 2 -- with 1 the host only
        began batch i + 2 once batch i had finished, its 0.7 ms of packing then landed inside decode(i + 1), and the
        encoder of batch i + 2 started where that decode ended instead of beside it: profiles/r04/evaluate_chain.txt).
        This is the loop of cli.py:449-495."""
        from collections import deque
        pending = deque()
        lookahead = max(1, min(int(lookahead), 2))           # the landing buffers / decoder slots rotate over 4 / 3
        for images, targets in batches:
            pending.append(self._evaluate_launch(images, targets, max_length, side_upload=True, coresident=True))
            if len(pending) > lookahead:
                yield self._evaluate_finish(pending.popleft())
        while pending:
            yield self._evaluate_finish(pending.popleft())

    def _evaluate_launch(self, images, targets: torch.Tensor, max_length: Optional[int], side_upload: bool = False,
                         rows_per_workgroup: int = 0, coresident: bool = False):
        """Enqueues the whole chain of one batch and the copy of its statistics to pinned memory; no host wait.
        ``coresident``: two streams -- everything up to the decoder's prepare on the first, the decode and what follows on the
        second (see evaluate_stream); otherwise the whole chain on the current stream."""
        from . import metrics as M
        import contextlib
        L = _lib.lib()
        tk = self.tokenizer
        raw = images
        # host-side preparation and everything that touches the CALLER's stream first (targets, the small constant tensors)
        T = int(max_length if max_length is not None else tk.max_sequence_length)
        tgt = targets.to(self.device, dtype=torch.int32).contiguous()
        if tgt.dim() != 2:
            raise RuntimeError(f"targets must be (B, L) token ids, got {tuple(tgt.shape)}")
        special = getattr(tk, "special_tokens", DEFAULT_SPECIAL_TOKENS)
        drop = sorted({tk.token_to_id[t] for t in special.values()}) if hasattr(tk, "token_to_id") else \
            sorted({tk.pad_token_id, tk.start_token_id, tk.end_token_id})
        if len(drop) > 8:
            raise NotImplementedError("img2latex_amd: at most 8 special token ids")
        dev = tgt.device
        key = (tuple(drop), tk.pad_token_id, dev)
        if getattr(self, "_drop_key", None) != key:
            self._drop = torch.tensor(drop, dtype=torch.int32, device=dev)
            self._pad = torch.tensor([tk.pad_token_id], dtype=torch.int32, device=dev)
            self._drop_key = key
        cur = torch.cuda.current_stream(self.device)
        if coresident:
            st = self.__dict__.setdefault("_ev_streams", {})
            if "enc" not in st:
                st["enc"] = torch.cuda.Stream(self.device)
                st["dec"] = torch.cuda.Stream(self.device, priority=-1)
                # (r04, measured and dropped: id compaction + statistics on a third stream, so that the decode stream goes from
                # decode(i) straight into decode(i + 1) -- three kernels sharing the CUs: 1.25 -> 1.63 ms per batch)
                st["stat"] = st["dec"]
                st["resident"], st["n"] = torch.zeros(1, dtype=torch.int32, device=self.device), 0
            s_enc, s_dec, s_stat = st["enc"], st["dec"], st["stat"]
            s_enc.wait_stream(cur)                           # the caller's tensors, tgt, the constants
            on_enc = lambda: torch.cuda.stream(s_enc)
            on_dec = lambda: torch.cuda.stream(s_dec)
            on_stat = lambda: torch.cuda.stream(s_stat)
        else:
            on_enc = on_dec = on_stat = contextlib.nullcontext
        with on_enc():
            if coresident and st["n"] > 0:                   # not before the previous batch's decode owns its compute units
                _lib.check(L.i2l_stream_wait_value32(st["resident"].data_ptr(), st["n"], 20000.0, _lib.stream_ptr()),
                           "stream_wait_value32")
            if not isinstance(images, torch.Tensor):
                from ..data import preprocess_batch
                enc_mod = self.model.encoder
                if side_upload and getattr(self, "_up_stream", None) is None:
                    self._up_stream = torch.cuda.Stream(self.device)
                with torch.cuda.device(self.device):
                    images = preprocess_batch(list(images), (enc_mod.img_height, enc_mod.img_width), enc_mod.channels, True,
                                              upload_stream=self._up_stream if side_upload else None)
            x = self._as_batch(images)
        B = x.shape[0]
        if tgt.shape[0] != B:
            raise RuntimeError(f"targets must be ({B}, L) token ids, got {tuple(tgt.shape)}")
        W = max(T, tgt.shape[1])
        prepared = None
        with on_enc(), torch.cuda.device(dev), torch.no_grad():
            t_ids = torch.zeros((B, W), dtype=torch.int32, device=dev)
            t_len = torch.empty((B,), dtype=torch.int32, device=dev)
            _lib.check(L.i2l_compact_ids(tgt.data_ptr(), B, tgt.shape[1], tgt.stride(0), -1, self._pad.data_ptr(), 1,
                                         t_ids.data_ptr(), W, t_len.data_ptr(), _lib.stream_ptr()), "compact_ids")
            enc = self.model.encoder(x)
            if coresident:
                w_, keep_, enc_c = self.model.decoder.prepare(enc, slot=("eval", id(self), st["n"] % 3))
                st["n"] += 1
                prepared = (w_, keep_, enc_c, self.model.decoder._ws)
                enc_done = torch.cuda.Event()
                enc_done.record(s_enc)
        with on_dec(), torch.cuda.device(dev):
            if coresident:
                s_dec.wait_event(enc_done)
                for t_ in (x, tgt, t_ids, t_len, enc):
                    t_.record_stream(s_dec)                  # allocated on the first stream, read on the second
            with torch.no_grad():
                ids, _ = self.model.greedy_ids(enc, tk.start_token_id, tk.end_token_id, T, stop=_lib.STOP_STICKY,
                                               select=_lib.SELECT_SOFTMAX, rows_per_workgroup=rows_per_workgroup,
                                               flags=_lib.FLAG_DECODE_GROUP16 if coresident else 0, prepared=prepared,
                                               resident=(st["resident"], st["n"]) if coresident else None)
            if coresident:
                decoded = torch.cuda.Event()
                decoded.record(s_dec)
        with on_stat(), torch.cuda.device(dev):
            if coresident:
                s_stat.wait_event(decoded)
                for t_ in (ids, t_ids, t_len):
                    t_.record_stream(s_stat)
            p_ids = torch.zeros((B, W), dtype=torch.int32, device=dev)
            p_len = torch.empty((B,), dtype=torch.int32, device=dev)
            with torch.no_grad():
                _lib.check(L.i2l_compact_ids(ids.data_ptr(), B, T, ids.stride(0), int(tk.end_token_id),
                                             self._drop.data_ptr(), len(drop), p_ids.data_ptr(), W, p_len.data_ptr(),
                                             _lib.stream_ptr()), "compact_ids")
                packed = M.device_sequence_statistics(p_ids, p_len, t_ids, t_len, 4, tk.pad_token_id, _max_len=W, defer=True)
            # statistics + the first id of every row (-3 = the grouped decode timed out) in ONE device->host copy
            packed = torch.cat([packed, ids[:, :1]], dim=1)
            # pinned landing buffers are reused (registering pinned memory costs ~1 ms): a ring of 4, more than the
            # two batches evaluate_stream ever has in flight
            ring = self.__dict__.setdefault("_host_ring", {"bufs": [None] * 4, "next": 0})
            slot = ring["next"]
            ring["next"] = (slot + 1) % 4
            if ring["bufs"][slot] is None or ring["bufs"][slot].numel() < packed.numel():
                ring["bufs"][slot] = torch.empty((packed.numel(),), dtype=torch.int32).pin_memory()
            host = ring["bufs"][slot][:packed.numel()].view(packed.shape)
            host.copy_(packed, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        return dict(host=host, done=done, keep=(packed, x, tgt), p_ids=p_ids, p_len=p_len, raw=(raw, targets, max_length),
                    rows_per_workgroup=rows_per_workgroup)

    def _evaluate_finish(self, h) -> Dict:
        from . import metrics as M
        h["done"].synchronize()
        host = h["host"]
        if _lib.ids_timed_out(host[:, 9:10]) and h["rows_per_workgroup"] == 0:
            # the timeout fallback of Seq2SeqModel.greedy_ids_host: the row-per-workgroup kernel needs no partner
            raw, targets, max_length = h["raw"]
            return self._evaluate_finish(self._evaluate_launch(raw, targets, max_length, rows_per_workgroup=1))
        _lib.check_ids(host[:, 9:10])
        out = M.metrics_from_packed(host)
        out["pred_ids"], out["pred_len"] = h["p_ids"], h["p_len"]
        return out

    def predict_batch(self, images, beam_size: int = 0, max_length: int = 141, temperature: float = 1.0,
                      top_k: int = 0, top_p: float = 0.0, batch_size: int = 16, seed: Optional[int] = None) -> List[str]:
        """predictor.py:205-394: every image through ``_prepare_image`` (:242), stacked, the batched greedy loop.  As in
        the reference beam_size is clamped to 0 (:231-235)."""
        if isinstance(images, torch.Tensor):
            images = [images] if images.dim() <= 3 else list(images)
        results: List[str] = []
        start, end = self.tokenizer.start_token_id, self.tokenizer.end_token_id
        for i in range(0, len(images), batch_size):
            x = torch.stack([self._prepare_image(im) for im in images[i:i + batch_size]])
            if x.dim() == 5 and x.shape[1] == 1:                            # :254-257
                x = x.squeeze(1)
            for seq in self.predict_batch_ids(x, max_length, temperature, top_k, top_p,
                                              None if seed is None else seed + i):
                if seq and seq[0] == start:                                 # :384-388
                    seq = seq[1:]
                if seq and seq[-1] == end:
                    seq = seq[:-1]
                results.append(self.tokenizer.decode(seq))
        return results

    def predict(self, image, beam_size: int = 0, max_length: int = 141, temperature: float = 1.0, top_k: int = 0,
                top_p: float = 0.0) -> str:
        """predictor.py:139-203: one image (path / PIL / ndarray / tensor) through ``_prepare_image`` and
        Seq2SeqModel.inference (greedy; beam clamped to 0)."""
        x = self._prepare_image(image).to(self.device)
        start, end = self.tokenizer.start_token_id, self.tokenizer.end_token_id
        with torch.no_grad():
            seq = self.model.inference(image=x, start_token_id=start, end_token_id=end, max_length=max_length,
                                       beam_size=0, temperature=temperature, top_k=top_k, top_p=top_p)
        if seq and isinstance(seq[0], list):
            seq = seq[0]
        if seq and seq[0] == start:
            seq = seq[1:]
        if seq and seq[-1] == end:
            seq = seq[:-1]
        return self.tokenizer.decode(seq)
