"""CPU restatement of the reference's image preprocessing -- TEST INFRASTRUCTURE (tests/, smoke(), cpu_baseline only).

Path: img2latex/data/utils.py:18-90 `load_image` (after PIL has decoded the file) =
  mode conversion (utils.py:45-49) -> `ResizeWithAspectRatio` (data/transforms.py:26-56: LANCZOS resize to the
  target height keeping the aspect ratio, then right-pad with white or centre-crop to the target width) ->
  float / 255 -> [-1,1] (1 channel) or ImageNet mean/std (3 channels) (utils.py:58-80).

The resize itself lives in a third-party dependency of the reference: Pillow (pinned here: 12.2.0, the version in
this image; `Image.resize(..., Resampling.LANCZOS)` -> libImaging/Resample.c).  Its published algorithm for 8-bit
images is restated below: separable two-pass convolution, horizontal pass first into an 8-bit intermediate image,
double-precision Lanczos-3 coefficients normalised per output pixel and rounded to 22-bit fixed point, accumulation
in int32 from 2^21, arithmetic shift, clamp to [0,255].  Pinned: tests/golden/preprocess.npz holds the outputs of
the real `load_image` (reference + Pillow) for generated images; tests/test_preprocess.py checks bit equality.
"""
import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2            # Resample.c
LANCZOS_SUPPORT = 3.0


def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


def _bicubic(x: float) -> float:
    """Resample.c bicubic_filter (Keys, a = -0.5): Image.resize()'s default filter (predictor.py:439)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


FILTERS = {"lanczos": (_lanczos, LANCZOS_SUPPORT), "bicubic": (_bicubic, 2.0)}


def precompute_coeffs(in_size: int, in0: float, in1: float, out_size: int, flt: str = "lanczos"):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc: (ksize, bounds (out,2) int, kk (out,ksize) int32)."""
    _filter, _support = FILTERS[flt]
    scale = float(np.float32(in1) - np.float32(in0)) / out_size
    filterscale = max(scale, 1.0)
    support = _support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        for x, w in enumerate(k):
            v = w * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_lanczos_u8(img: np.ndarray, out_w: int, out_h: int, flt: str = "lanczos") -> np.ndarray:
    """Image.resize((out_w, out_h), LANCZOS) for mode L (H,W) or RGB (H,W,3) uint8 (Image.py resize + Resample.c);
    flt="bicubic" = Image.resize((out_w, out_h)) with no filter named."""
    h, w = img.shape[:2]
    if (w, h) == (out_w, out_h):
        return img.copy()
    src = img.reshape(h, w, -1).astype(np.int64)
    need_h, need_v = out_w != w, out_h != h
    _, bh, kh = precompute_coeffs(w, 0.0, float(w), out_w, flt)
    _, bv, kv = precompute_coeffs(h, 0.0, float(h), out_h, flt)
    ybox_first = int(bv[0, 0])
    ybox_last = int(bv[out_h - 1, 0] + bv[out_h - 1, 1])
    cur = src
    if need_h:
        bv = bv.copy()
        bv[:, 0] -= ybox_first
        rows = cur[ybox_first:ybox_last]
        tmp = np.zeros((rows.shape[0], out_w, src.shape[2]), np.int64)
        for xx in range(out_w):
            x0, n = int(bh[xx, 0]), int(bh[xx, 1])
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(rows[:, x0:x0 + n, :], kh[xx, :n].astype(np.int64), axes=([1], [0]))
            tmp[:, xx, :] = _clip8(acc)
        cur = tmp
    if need_v:
        out = np.zeros((out_h, cur.shape[1], src.shape[2]), np.int64)
        for yy in range(out_h):
            y0, n = int(bv[yy, 0]), int(bv[yy, 1])
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kv[yy, :n].astype(np.int64), cur[y0:y0 + n], axes=([0], [0]))
            out[yy] = _clip8(acc)
        cur = out
    return cur.astype(np.uint8).reshape((out_h, out_w) + img.shape[2:])


def convert_mode(img: np.ndarray, channels: int) -> np.ndarray:
    """utils.py:45-49 `img.convert("L" / "RGB")` for uint8 L (H,W) / RGB (H,W,3) sources (Pillow Convert.c rgb2l / l2rgb)."""
    if channels == 1 and img.ndim == 3:
        r, g, b = (img[..., i].astype(np.int64) for i in range(3))
        return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)
    if channels == 3 and img.ndim == 2:
        return np.repeat(img[..., None], 3, axis=2)
    return img


def resize_with_aspect_ratio(img: np.ndarray, target_h: int, target_w: int) -> np.ndarray:
    """transforms.py:26-56 on a uint8 array."""
    h, w = img.shape[:2]
    new_w = int(round(target_h * (w / h)))
    res = resize_lanczos_u8(img, new_w, target_h)
    if new_w == target_w:
        return res
    if new_w < target_w:
        # Image.new(mode, size, 255): for "RGB" the integer colour 255 is the packed pixel 0x000000FF = (255, 0, 0),
        # so 3-channel images are padded with RED, not white (transforms.py:44-47 as executed by Pillow)
        out = np.zeros((target_h, target_w) + img.shape[2:], np.uint8)
        out[..., 0] = 255
        if img.ndim == 2:
            out[...] = 255
        out[:, :new_w] = res
        return out
    left = (new_w - target_w) // 2
    return res[:, left:left + target_w]


def load_image_from_array(img: np.ndarray, img_size: Tuple[int, int], channels: int, normalize: bool = True) -> np.ndarray:
    """utils.py:37-80 given the decoded uint8 image: (C, H, W) float32."""
    img = convert_mode(img, channels)
    img = resize_with_aspect_ratio(img, img_size[0], img_size[1])
    if channels == 1:
        t = img[None].astype(np.float32)
    else:
        t = np.transpose(img, (2, 0, 1)).astype(np.float32)
    t = t / np.float32(255.0)
    if normalize:
        if channels == 1:
            t = t * np.float32(2.0) - np.float32(1.0)
        else:
            mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(-1, 1, 1)
            std = np.array([0.229, 0.224, 0.225], np.float32).reshape(-1, 1, 1)
            t = (t - mean) / std
    return t.astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------
# Predictor._prepare_image (img2latex/training/predictor.py:396-521): every input type -> (1, C, 64, 800)
# ---------------------------------------------------------------------------------------------------------------
def prepare_image(image, model_type: str = "cnn_lstm"):
    """predictor.py:396-462.  str: load_image (:418-420; a missing / unreadable file -> zeros, utils.py:84-90);
    tensor (:421-423) / ndarray (:424-427): _preprocess_tensor; PIL image (:428-451): convert, resize((800, 64)) with
    Pillow's default BICUBIC, /255, *2-1 on every channel.  Returns a float32 torch tensor (1, C, 64, 800)."""
    import torch
    from PIL import Image
    img_size, channels = (64, 800), (1 if model_type == "cnn_lstm" else 3)
    if isinstance(image, str):
        try:
            pil = Image.open(image)
            want = "L" if channels == 1 else "RGB"
            arr = np.array(pil if pil.mode in ("L", "RGB") else pil.convert(want))
            t = torch.from_numpy(load_image_from_array(arr, img_size, channels, True))
        except Exception:
            t = torch.zeros((channels, img_size[0], img_size[1]))
    elif isinstance(image, torch.Tensor):
        t = preprocess_tensor(image, img_size)
    elif isinstance(image, np.ndarray):
        t = preprocess_tensor(numpy_to_tensor(image), img_size)
    elif isinstance(image, Image.Image):
        want = "L" if channels == 1 else "RGB"
        arr = np.array(image if image.mode in ("L", "RGB") else image.convert(want))
        arr = resize_lanczos_u8(convert_mode(arr, channels), img_size[1], img_size[0], "bicubic")
        arr = arr[None] if channels == 1 else np.transpose(arr, (2, 0, 1))
        t = torch.from_numpy(arr.copy()).float() / 255.0
        t = t * 2.0 - 1.0
    else:
        raise TypeError(f"Unsupported image type: {type(image)}")
    if model_type == "resnet_lstm" and t.shape[0] == 1:
        t = t.repeat(3, 1, 1)
    return t.unsqueeze(0) if t.dim() == 3 else t


def preprocess_tensor(tensor, img_size):
    """predictor.py:464-499."""
    import torch
    if tensor.dim() == 2:
        tensor = tensor.unsqueeze(0)
    if tuple(tensor.shape[-2:]) != tuple(img_size):
        tensor = torch.nn.functional.interpolate(tensor.unsqueeze(0) if tensor.dim() == 3 else tensor, size=img_size,
                                                 mode="bilinear", align_corners=False)
        if tensor.dim() == 4 and tensor.shape[0] == 1:
            tensor = tensor.squeeze(0)
    if tensor.min() < 0 or tensor.max() > 1:
        tensor = tensor / 255.0
        tensor = tensor * 2.0 - 1.0
    return tensor


def numpy_to_tensor(array: np.ndarray):
    """predictor.py:501-521."""
    import torch
    if array.ndim == 2:
        array = np.expand_dims(array, axis=0)
    elif array.ndim == 3 and array.shape[0] not in [1, 3]:
        array = np.transpose(array, (2, 0, 1))
    return torch.from_numpy(array).float()
