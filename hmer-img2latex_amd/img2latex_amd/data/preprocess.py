"""Image preprocessing on the device with the reference's results (img2latex/data/utils.py:18-110,
data/transforms.py:26-56): mode conversion, aspect-preserving LANCZOS resize to the target height, right-pad /
centre-crop to the target width, /255 and normalisation -- for a whole ragged batch in one pair of kernel launches,
bit-identical to `load_image` (Pillow 12.2 semantics; golden vectors in tests/golden/preprocess.npz).

File decoding stays with PIL on the host (it is I/O + entropy decoding, not part of the hot path); everything after
`Image.open` runs in HIP.  The Lanczos weights are built by the library's host helper (`i2l_lanczos_coeffs`: double
precision + libm, the arithmetic Pillow itself runs) and cached per (source size, target size).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch

from .. import _lib


# struct i2l_resize_plan (include/img2latex_hip.h) as a numpy record: a whole batch of plans is ONE array
PLAN_DTYPE = np.dtype([("src_offset", "<i8"), ("tmp_offset", "<i8"), ("bh_offset", "<i8"), ("kh_offset", "<i8"),
                       ("bv_offset", "<i8"), ("kv_offset", "<i8"), ("src_h", "<i4"), ("src_w", "<i4"), ("src_c", "<i4"),
                       ("new_w", "<i4"), ("ybox_first", "<i4"), ("tmp_rows", "<i4"), ("need_h", "<i4"), ("need_v", "<i4"),
                       ("kh_ksize", "<i4"), ("kv_ksize", "<i4")])
assert PLAN_DTYPE.itemsize == 88


class _TablePool:
    """Resampling tables resident on the device, one per (source size, target size, filter), shared by every image
    and every batch that needs them: a batch only pays (threaded, in the library's host helper) for sizes it has not
    seen before, and uploads nothing else but its pixels and plans.  Grow-only up to `capacity` int32 elements, then
    reset.  One pool per device."""

    def __init__(self, device: torch.device, capacity: int = 16 << 20):
        self.device, self.capacity = device, capacity
        self.tables = torch.empty((capacity,), dtype=torch.int32, device=device)
        self.used = 0
        self.index = {}            # (in, out, filter) -> (bounds offset, weights offset, ksize, first, last)

    def lookup(self, keys):
        """Makes every key resident; returns nothing (read self.index)."""
        missing = [k for k in dict.fromkeys(keys) if k not in self.index]
        if not missing:
            return
        L = _lib.lib()
        ks = [L.i2l_resample_ksize(f, a, b) for (a, b, f) in missing]
        if min(ks) <= 0:
            raise ValueError("cannot resample an empty axis")
        sizes = [b * (2 + k) for (a, b, f), k in zip(missing, ks)]
        total = int(sum(sizes))
        if total > self.capacity:
            raise ValueError("resampling tables of one batch exceed the table pool")
        if self.used + total > self.capacity:                     # full: start over (tables of earlier batches are
            torch.cuda.current_stream(self.device).synchronize()  # dropped once nothing in flight reads them)
            self.index.clear()
            self.used = 0
            return self.lookup(keys)
        offs = np.zeros(len(missing), np.int64)
        offs[1:] = np.cumsum(sizes[:-1])
        host = torch.empty((total,), dtype=torch.int32).pin_memory()
        for flt in sorted({f for (_, _, f) in missing}):
            sel = [i for i, (_, _, f) in enumerate(missing) if f == flt]
            ins = np.array([missing[i][0] for i in sel], np.int32)
            outs = np.array([missing[i][1] for i in sel], np.int32)
            o = np.ascontiguousarray(offs[sel])
            _lib.check(L.i2l_resample_coeffs_batch(flt, len(sel), ins.ctypes.data, outs.ctypes.data, o.ctypes.data,
                                                   host.data_ptr(), min(16, os.cpu_count() or 1)), "resample_coeffs_batch")
        hv = host.numpy()
        for (key, k, off) in zip(missing, ks, offs.tolist()):
            b = key[1]
            first = int(hv[off])
            last = int(hv[off + 2 * (b - 1)] + hv[off + 2 * (b - 1) + 1])
            self.index[key] = (self.used + off, self.used + off + 2 * b, k, first, last)
        self.tables[self.used:self.used + total].copy_(host, non_blocking=True)
        # the pinned staging block must outlive the asynchronous copy
        torch.cuda.current_stream(self.device).synchronize()
        self.used += total


_POOLS = {}
_STAGING = {}


def _pool(dev: torch.device) -> _TablePool:
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _POOLS:
        _POOLS[key] = _TablePool(dev)
    return _POOLS[key]


def _staging(dev: torch.device, nbytes: int):
    """Two pinned uint8 blocks per device, used alternately; a block is reused only after the copy that last read it
    has finished (its event)."""
    key = (dev.type, dev.index)
    st = _STAGING.setdefault(key, {"bufs": [None, None], "events": [None, None], "next": 0})
    i = st["next"]
    st["next"] = i ^ 1
    if st["events"][i] is not None:
        st["events"][i].synchronize()
    if st["bufs"][i] is None or st["bufs"][i].numel() < nbytes:
        st["bufs"][i] = torch.empty((max(nbytes, 1 << 20) * 5 // 4,), dtype=torch.uint8).pin_memory()
    return st, i, st["bufs"][i]


def preprocess_batch(images: Sequence[np.ndarray], img_size: Tuple[int, int] = (64, 800), channels: int = 1,
                     normalize=True, device=None, keep_aspect: bool = True, resample: str = "lanczos",
                     upload_stream=None) -> torch.Tensor:
    """`load_image` for already decoded images: uint8 arrays (H, W) ["L"] or (H, W, 3) ["RGB"], any sizes.
    Returns (n, channels, img_size[0], img_size[1]) float32 on the device.

    ``keep_aspect=False, resample="bicubic", normalize="symmetric"`` is the PIL.Image branch of the reference's
    ``Predictor._prepare_image`` instead (predictor.py:432-451): ``image.resize((W, H))`` with Pillow's default filter
    straight to the target size, then ``x / 255 * 2 - 1`` on every channel.

    Host work per batch: one pass over the shapes, table look-ups in the device-resident pool (new sizes are computed
    by i2l_resample_coeffs_batch on the host's cores), ONE packed copy of the pixels into pinned memory and two
    asynchronous upload (pixels + plans).  ``upload_stream``: run that upload on a side stream (the kernels, on the
    current stream, wait for it) so that it overlaps whatever the current stream is still doing."""
    flt = {"lanczos": _lib.FILTER_LANCZOS, "bicubic": _lib.FILTER_BICUBIC}[resample]
    normalize = 2 if normalize == "symmetric" else int(bool(normalize))
    if channels not in (1, 3):
        raise ValueError("channels must be 1 or 3")
    if not torch.cuda.is_available():
        raise RuntimeError("img2latex_amd: the preprocessing kernels need the ROCm device; there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    out_h, out_w = int(img_size[0]), int(img_size[1])
    n = len(images)
    if n == 0:
        return torch.empty((0, channels, out_h, out_w), dtype=torch.float32, device=dev)
    flats = []
    shapes = np.empty((n, 3), np.int64)
    for i, img in enumerate(images):
        if not isinstance(img, np.ndarray) or img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] != 3):
            raise TypeError("images must be uint8 arrays of shape (H, W) or (H, W, 3)")
        shapes[i, 0], shapes[i, 1], shapes[i, 2] = img.shape[0], img.shape[1], (1 if img.ndim == 2 else 3)
        flats.append(img.reshape(-1) if img.flags.c_contiguous else np.ascontiguousarray(img).reshape(-1))
    h, w, c = shapes[:, 0], shapes[:, 1], shapes[:, 2]
    if int(h.min()) == 0 or int(w.min()) == 0:
        raise ValueError("empty image")
    # transforms.py:33-36: int(round(target_h * (w / h))) -- Python's round() and numpy's both round half to even
    new_w = np.round(out_h * (w / h)).astype(np.int64) if keep_aspect else np.full(n, out_w, np.int64)
    if int(new_w.min()) <= 0:
        bad = int(np.argmin(new_w))
        raise ValueError(f"image {bad} ({int(h[bad])}x{int(w[bad])}) collapses to zero width at height {out_h}")
    pool = _pool(dev)
    hkeys = [(int(a), int(b), flt) for a, b in zip(w, new_w)]
    vkeys = [(int(a), out_h, flt) for a in h]
    with torch.cuda.device(dev):
        pool.lookup(hkeys + vkeys)
        ent_h = np.array([pool.index[k] for k in hkeys], np.int64)
        ent_v = np.array([pool.index[k] for k in vkeys], np.int64)
        plans = np.zeros(n, PLAN_DTYPE)
        sizes = h * w * c
        src_off = np.zeros(n, np.int64)
        src_off[1:] = np.cumsum(sizes[:-1])
        need_h = new_w != w
        tmp_rows = np.where(need_h, ent_v[:, 4] - ent_v[:, 3], 0)
        tmp_bytes = (tmp_rows * new_w * channels + 255) // 256 * 256
        tmp_off = np.zeros(n, np.int64)
        tmp_off[1:] = np.cumsum(tmp_bytes[:-1])
        plans["src_offset"], plans["tmp_offset"] = src_off, tmp_off
        plans["bh_offset"], plans["kh_offset"], plans["kh_ksize"] = ent_h[:, 0], ent_h[:, 1], ent_h[:, 2]
        plans["bv_offset"], plans["kv_offset"], plans["kv_ksize"] = ent_v[:, 0], ent_v[:, 1], ent_v[:, 2]
        plans["src_h"], plans["src_w"], plans["src_c"], plans["new_w"] = h, w, c, new_w
        plans["ybox_first"] = np.where(need_h, ent_v[:, 3], 0)
        plans["tmp_rows"] = tmp_rows
        plans["need_h"], plans["need_v"] = need_h, h != out_h
        total_px = int(sizes.sum())
        plan_bytes = n * PLAN_DTYPE.itemsize
        st, slot, pinned = _staging(dev, total_px + plan_bytes + 256)
        host = pinned.numpy()
        np.concatenate(flats, out=host[:total_px])
        p0 = (total_px + 255) // 256 * 256
        host[p0:p0 + plan_bytes] = plans.view(np.uint8)
        cur = torch.cuda.current_stream(dev)
        with torch.cuda.stream(upload_stream if upload_stream is not None else cur):
            # allocated under the uploading stream: memory the caching allocator hands out there has no pending work
            # of the compute stream on it, so the copy need not wait for the previous batch's kernels
            d_all = torch.empty((p0 + plan_bytes,), dtype=torch.uint8, device=dev)
            d_all.copy_(pinned[:p0 + plan_bytes], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        st["events"][slot] = ev
        if upload_stream is not None:
            cur.wait_event(ev)
            d_all.record_stream(cur)
        max_tmp_px = int((tmp_rows * new_w).max())
        ws = torch.empty((max(int(tmp_bytes.sum()), 16),), dtype=torch.uint8, device=dev)
        out = torch.empty((n, channels, out_h, out_w), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().i2l_preprocess_images(d_all.data_ptr(), d_all.data_ptr() + p0, pool.tables.data_ptr(), n,
                                                    max_tmp_px, channels, out_h, out_w, normalize,
                                                    ws.data_ptr(), out.data_ptr(), _lib.stream_ptr()), "preprocess_images")
    return out


def load_image(image_path: str, img_size: Tuple[int, int] = (64, 800), channels: int = 1,
               normalize: bool = True) -> torch.Tensor:
    """utils.py:18-90 with the reference's signature and defaults.  PIL only opens / decodes the file; the tensor
    (channels, H, W) is produced on the device (the reference returns the same values on the CPU).  Like the
    reference, any failure yields a zero image."""
    try:
        from PIL import Image
        img = Image.open(image_path)
        if channels == 1:
            arr = np.array(img if img.mode in ("L", "RGB") else img.convert("L"))
        else:
            arr = np.array(img if img.mode in ("L", "RGB") else img.convert("RGB"))
        return preprocess_batch([arr], img_size, channels, normalize)[0]
    except Exception:                                             # utils.py:84-90
        dev = torch.device("cuda", torch.cuda.current_device())
        return torch.zeros((channels, img_size[0], img_size[1]), device=dev)


def resize_bilinear(t: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """``torch.nn.functional.interpolate(t, size, mode="bilinear", align_corners=False)`` of a (..., H, W) fp32 device
    tensor on the HIP kernel (the tensor branch of Predictor._prepare_image, predictor.py:483-491)."""
    if not t.is_cuda:
        raise RuntimeError("img2latex_amd: resize_bilinear needs a tensor on the ROCm device; there is no CPU fallback")
    t = t.to(torch.float32).contiguous()
    out = torch.empty(t.shape[:-2] + (int(size[0]), int(size[1])), dtype=torch.float32, device=t.device)
    planes = int(np.prod(t.shape[:-2])) if t.dim() > 2 else 1
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().i2l_resize_bilinear_f32(t.data_ptr(), out.data_ptr(), planes, t.shape[-2], t.shape[-1],
                                                      int(size[0]), int(size[1]), _lib.stream_ptr()), "resize_bilinear_f32")
    return out


def batch_convert_for_resnet(batch_tensor: torch.Tensor) -> torch.Tensor:
    """utils.py:93-110: gray -> RGB by repeating the channel."""
    if batch_tensor.shape[1] == 3:
        return batch_tensor
    return batch_tensor.repeat(1, 3, 1, 1)
