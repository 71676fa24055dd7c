"""Image preprocessing on the device with the reference's results (img2latex/data/utils.py:18-110,
data/transforms.py:26-56): mode conversion, aspect-preserving LANCZOS resize to the target height, right-pad /
centre-crop to the target width, /255 and normalisation -- for a whole ragged batch in one pair of kernel launches,
bit-identical to `load_image` (Pillow 12.2 semantics; golden vectors in tests/golden/preprocess.npz).

File decoding stays with PIL on the host (it is I/O + entropy decoding, not part of the hot path); everything after
`Image.open` runs in HIP.  Since r04 that includes the resampling tables: `i2l_resample_coeffs_device` builds the
Lanczos / bicubic weights of a whole batch on the device from its size list (the host's double-precision arithmetic
repeated operation by operation; `tables="host"` keeps the library's host helper -- libm, the arithmetic Pillow itself
runs -- with its device-resident cache), and the ragged pages are gathered into the pinned upload block by
`i2l_pack_host` on several host threads.  Per batch the host now walks the shapes, fills one record array and uploads.
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


def _ksizes(flt: int, ins: np.ndarray, outs: np.ndarray) -> np.ndarray:
    """i2l_resample_ksize for arrays (preprocess.hip: (int)ceil(support * max(1, (double)(float)in / out)) * 2 + 1)."""
    fs = np.maximum(ins.astype(np.float32).astype(np.float64) / outs, 1.0)
    return np.ceil((2.0 if flt == _lib.FILTER_BICUBIC else 3.0) * fs).astype(np.int64) * 2 + 1


def _extents(flt: int, ins: np.ndarray, outs: np.ndarray):
    """First source sample of output 0 and one past the last source sample of output out - 1 (what the vertical pass
    needs of the horizontal one): the bounds arithmetic of i2l_resample_coeffs for those two outputs, in the same doubles."""
    scale = ins.astype(np.float32).astype(np.float64) / outs
    support = (2.0 if flt == _lib.FILTER_BICUBIC else 3.0) * np.maximum(scale, 1.0)
    c0 = (0 + 0.5) * scale
    first = np.maximum(np.trunc(c0 - support + 0.5), 0.0)
    cl = (outs - 1 + 0.5) * scale
    last = np.minimum(np.trunc(cl + support + 0.5), ins.astype(np.float64))
    return first.astype(np.int64), last.astype(np.int64)


def preprocess_batch(images: Sequence[np.ndarray], img_size: Tuple[int, int] = (64, 800), channels: int = 1,
                     normalize=True, device=None, keep_aspect: bool = True, resample: str = "lanczos",
                     upload_stream=None, tables: str = "device") -> torch.Tensor:
    """`load_image` for already decoded images: uint8 arrays (H, W) ["L"] or (H, W, 3) ["RGB"], any sizes.
    Returns (n, channels, img_size[0], img_size[1]) float32 on the device.

    ``keep_aspect=False, resample="bicubic", normalize="symmetric"`` is the PIL.Image branch of the reference's
    ``Predictor._prepare_image`` instead (predictor.py:432-451): ``image.resize((W, H))`` with Pillow's default filter
    straight to the target size, then ``x / 255 * 2 - 1`` on every channel.

    Host work per batch: one pass over the shapes, one record array of plans, ONE packed copy of the pixels into pinned
    memory (i2l_pack_host, several threads) and one asynchronous upload (pixels + plans + size list); the resampling
    tables are built on the device (``tables="device"``, i2l_resample_coeffs_device).  ``tables="host"``: the library's
    host helper (libm: Pillow's own arithmetic) with look-ups in a device-resident pool, new sizes computed on the host's
    cores.  ``upload_stream``: run the upload on a side stream (the kernels, on the current stream, wait for it) so that
    it overlaps whatever the current stream is still doing."""
    if tables not in ("device", "host"):
        raise ValueError("tables must be 'device' or 'host'")
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
    # one pass over the Python objects, as little per image as the interpreter allows (256 pages: ~0.15 ms)
    u8 = np.dtype(np.uint8)
    try:
        ok = all(a.dtype is u8 or a.dtype == u8 for a in images)
        dims = [a.shape if a.ndim == 3 else a.shape + (1,) for a in images]
    except AttributeError:
        ok = False
    if not ok or any(len(d) != 3 or (d[2] != 3 and d[2] != 1) for d in dims) or any(a.ndim == 3 and a.shape[2] == 1 for a in images):
        raise TypeError("images must be uint8 arrays of shape (H, W) or (H, W, 3)")
    flats = [a if a.flags.c_contiguous else np.ascontiguousarray(a) for a in images]
    shapes = np.array(dims, np.int64).reshape(n, 3)
    h, w, c = shapes[:, 0], shapes[:, 1], shapes[:, 2]
    if int(h.min()) == 0 or int(w.min()) == 0:
        raise ValueError("empty image")
    # transforms.py:33-36: int(round(target_h * (w / h))) -- Python's round() and numpy's both round half to even
    new_w = np.round(out_h * (w / h)).astype(np.int64) if keep_aspect else np.full(n, out_w, np.int64)
    if int(new_w.min()) <= 0:
        bad = int(np.argmin(new_w))
        raise ValueError(f"image {bad} ({int(h[bad])}x{int(w[bad])}) collapses to zero width at height {out_h}")
    with torch.cuda.device(dev):
        if tables == "host":
            pool = _pool(dev)
            hkeys = [(int(a), int(b), flt) for a, b in zip(w, new_w)]
            vkeys = [(int(a), out_h, flt) for a in h]
            pool.lookup(hkeys + vkeys)
            ent_h = np.array([pool.index[k] for k in hkeys], np.int64)
            ent_v = np.array([pool.index[k] for k in vkeys], np.int64)
            tab_req = None
        else:
            # 2n tables (n horizontal w -> new_w, n vertical h -> out_h) laid out back to back in one per-batch buffer
            t_in = np.concatenate([w, h]).astype(np.int32)
            t_out = np.concatenate([new_w, np.full(n, out_h, np.int64)]).astype(np.int32)
            ks = _ksizes(flt, t_in, t_out)
            if int(ks.max()) > 2 * 512:
                raise ValueError("down-scaling by more than ~80x is not supported")
            t_size = t_out.astype(np.int64) * (2 + ks)
            t_off = np.zeros(2 * n, np.int64)
            t_off[1:] = np.cumsum(t_size[:-1])
            first, last = _extents(flt, t_in[n:], t_out[n:])
            ent_h = np.stack([t_off[:n], t_off[:n] + 2 * t_out[:n], ks[:n]], axis=1)
            ent_v = np.stack([t_off[n:], t_off[n:] + 2 * t_out[n:], ks[n:], first, last], axis=1)
            tab_req = (t_in, t_out, t_off, int(t_size.sum()), int(t_out.max()))
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
        req_bytes = 0 if tab_req is None else 2 * n * (4 + 4 + 8)
        p0 = (total_px + 255) // 256 * 256
        p1 = (p0 + plan_bytes + 255) // 256 * 256
        st, slot, pinned = _staging(dev, p1 + req_bytes + 256)
        host = pinned.numpy()
        # the pages -> one pinned block, on several host threads (one thread copies 16 MB in ~3 ms)
        ptrs = np.fromiter((a.__array_interface__["data"][0] for a in flats), dtype=np.uint64, count=n)
        _lib.check(_lib.lib().i2l_pack_host(ptrs.ctypes.data, sizes.ctypes.data, src_off.ctypes.data, n, pinned.data_ptr(),
                                            min(4, os.cpu_count() or 1)), "pack_host")      # 16 MB: 0.39 / 0.18 / 0.25 ms on 1 / 4 / 8 threads
        host[p0:p0 + plan_bytes] = plans.view(np.uint8)
        if tab_req is not None:
            t_in, t_out, t_off, t_total, t_max_out = tab_req
            host[p1:p1 + 8 * n] = t_in.view(np.uint8)
            host[p1 + 8 * n:p1 + 16 * n] = t_out.view(np.uint8)
            host[p1 + 16 * n:p1 + 32 * n] = t_off.view(np.uint8)
        cur = torch.cuda.current_stream(dev)
        with torch.cuda.stream(upload_stream if upload_stream is not None else cur):
            # allocated under the uploading stream: memory the caching allocator hands out there has no pending work
            # of the compute stream on it, so the copy need not wait for the previous batch's kernels
            d_all = torch.empty((p1 + req_bytes,), dtype=torch.uint8, device=dev)
            d_all.copy_(pinned[:p1 + req_bytes], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        st["events"][slot] = ev
        if upload_stream is not None:
            cur.wait_event(ev)
            d_all.record_stream(cur)
        max_tmp_px = int((tmp_rows * new_w).max())
        ws = torch.empty((max(int(tmp_bytes.sum()), 16),), dtype=torch.uint8, device=dev)
        out = torch.empty((n, channels, out_h, out_w), dtype=torch.float32, device=dev)
        if tab_req is None:
            tab = pool.tables
        else:
            tab = torch.empty((max(t_total, 4),), dtype=torch.int32, device=dev)
            _lib.check(_lib.lib().i2l_resample_coeffs_device(flt, 2 * n, d_all.data_ptr() + p1, d_all.data_ptr() + p1 + 8 * n,
                                                             d_all.data_ptr() + p1 + 16 * n, tab.data_ptr(), t_max_out,
                                                             _lib.stream_ptr()), "resample_coeffs_device")
        _lib.check(_lib.lib().i2l_preprocess_images(d_all.data_ptr(), d_all.data_ptr() + p0, tab.data_ptr(), n,
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
