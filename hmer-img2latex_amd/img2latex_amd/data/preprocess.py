"""Image preprocessing on the device with the reference's results (img2latex/data/utils.py:18-110,
data/transforms.py:26-56): mode conversion, aspect-preserving LANCZOS resize to the target height, right-pad /
centre-crop to the target width, /255 and normalisation -- for a whole ragged batch in one pair of kernel launches,
bit-identical to `load_image` (Pillow 12.2 semantics; golden vectors in tests/golden/preprocess.npz).

File decoding stays with PIL on the host (it is I/O + entropy decoding, not part of the hot path); everything after
`Image.open` runs in HIP.  The Lanczos weights are built by the library's host helper (`i2l_lanczos_coeffs`: double
precision + libm, the arithmetic Pillow itself runs) and cached per (source size, target size).
"""
from __future__ import annotations

import ctypes
import functools
from typing import List, Sequence, Tuple

import numpy as np
import torch

from .. import _lib


class ResizePlan(ctypes.Structure):
    """struct i2l_resize_plan (include/img2latex_hip.h)."""
    _fields_ = [("src_offset", ctypes.c_int64), ("tmp_offset", ctypes.c_int64),
                ("bh_offset", ctypes.c_int64), ("kh_offset", ctypes.c_int64),
                ("bv_offset", ctypes.c_int64), ("kv_offset", ctypes.c_int64),
                ("src_h", ctypes.c_int32), ("src_w", ctypes.c_int32), ("src_c", ctypes.c_int32),
                ("new_w", ctypes.c_int32), ("ybox_first", ctypes.c_int32), ("tmp_rows", ctypes.c_int32),
                ("need_h", ctypes.c_int32), ("need_v", ctypes.c_int32),
                ("kh_ksize", ctypes.c_int32), ("kv_ksize", ctypes.c_int32)]


@functools.lru_cache(maxsize=4096)
def _coeffs(in_size: int, out_size: int, flt: int = _lib.FILTER_LANCZOS) -> Tuple[int, np.ndarray, np.ndarray]:
    L = _lib.lib()
    ksize = L.i2l_resample_ksize(flt, in_size, out_size)
    if ksize <= 0:
        raise ValueError(f"cannot resample {in_size} -> {out_size}")
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    _lib.check(L.i2l_resample_coeffs(flt, in_size, out_size, bounds.ctypes.data, kk.ctypes.data), "resample_coeffs")
    return ksize, bounds, kk


def preprocess_batch(images: Sequence[np.ndarray], img_size: Tuple[int, int] = (64, 800), channels: int = 1,
                     normalize=True, device=None, keep_aspect: bool = True, resample: str = "lanczos") -> torch.Tensor:
    """`load_image` for already decoded images: uint8 arrays (H, W) ["L"] or (H, W, 3) ["RGB"], any sizes.
    Returns (n, channels, img_size[0], img_size[1]) float32 on the device.

    ``keep_aspect=False, resample="bicubic", normalize="symmetric"`` is the PIL.Image branch of the reference's
    ``Predictor._prepare_image`` instead (predictor.py:432-451): ``image.resize((W, H))`` with Pillow's default filter
    straight to the target size, then ``x / 255 * 2 - 1`` on every channel."""
    flt = {"lanczos": _lib.FILTER_LANCZOS, "bicubic": _lib.FILTER_BICUBIC}[resample]
    normalize = 2 if normalize == "symmetric" else int(bool(normalize))
    if channels not in (1, 3):
        raise ValueError("channels must be 1 or 3")
    if not torch.cuda.is_available():
        raise RuntimeError("img2latex_amd: the preprocessing kernels need the ROCm device; there is no CPU fallback")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out_h, out_w = int(img_size[0]), int(img_size[1])
    n = len(images)
    if n == 0:
        return torch.empty((0, channels, out_h, out_w), dtype=torch.float32, device=dev)
    plans = (ResizePlan * n)()
    tables: List[np.ndarray] = []
    pix: List[np.ndarray] = []
    t_off = p_off = w_off = 0
    max_tmp_px = 0
    for i, img in enumerate(images):
        img = np.ascontiguousarray(img)
        if img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] != 3):
            raise TypeError("images must be uint8 arrays of shape (H, W) or (H, W, 3)")
        h, w = img.shape[:2]
        if h == 0 or w == 0:
            raise ValueError("empty image")
        new_w = int(round(out_h * (w / h))) if keep_aspect else out_w   # transforms.py:33-36 / predictor.py:439
        if new_w <= 0:
            raise ValueError(f"image {i} ({h}x{w}) collapses to zero width at height {out_h}")
        p = plans[i]
        p.src_offset, p.src_h, p.src_w, p.src_c = p_off, h, w, 1 if img.ndim == 2 else 3
        p.new_w, p.need_h, p.need_v = new_w, int(new_w != w), int(out_h != h)
        kh_k, bh, kh = _coeffs(w, new_w, flt)
        kv_k, bv, kv = _coeffs(h, out_h, flt)
        ybox_first = int(bv[0, 0])
        ybox_last = int(bv[out_h - 1, 0] + bv[out_h - 1, 1])
        if p.need_h:                                              # Resample.c: shift bounds for the vertical pass
            bv = bv.copy()
            bv[:, 0] -= ybox_first
            p.ybox_first, p.tmp_rows = ybox_first, ybox_last - ybox_first
            p.tmp_offset = w_off
            w_off += (p.tmp_rows * new_w * channels + 255) // 256 * 256
            max_tmp_px = max(max_tmp_px, p.tmp_rows * new_w)
        p.kh_ksize, p.kv_ksize = kh_k, kv_k
        for arr, name in ((bh, "bh_offset"), (kh, "kh_offset"), (bv, "bv_offset"), (kv, "kv_offset")):
            setattr(p, name, t_off)
            tables.append(arr.reshape(-1))
            t_off += arr.size
        pix.append(img.reshape(-1))
        p_off += (img.size + 255) // 256 * 256
    pixels = np.zeros((p_off,), np.uint8)
    for p, a in zip(plans, pix):
        pixels[p.src_offset: p.src_offset + a.size] = a
    d_pixels = torch.from_numpy(pixels).to(dev)
    d_tables = torch.from_numpy(np.concatenate(tables)).to(dev)
    d_plans = torch.from_numpy(np.frombuffer(bytes(plans), dtype=np.uint8).copy()).to(dev)
    ws = torch.empty((max(w_off, 16),), dtype=torch.uint8, device=dev)
    out = torch.empty((n, channels, out_h, out_w), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().i2l_preprocess_images(d_pixels.data_ptr(), d_plans.data_ptr(), d_tables.data_ptr(), n,
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
