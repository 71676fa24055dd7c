"""The inputs of the predict-chain fixture (make_golden_predict.py hands them to the reference's Predictor,
tests/test_predict_chain.py hands the same objects to the drop-in): synthetic formula-like pages and one object per
input type `Predictor._prepare_image` accepts (predictor.py:396-462).  No reference import."""
import numpy as np
import torch
from PIL import Image

from img2latex_amd import synth


def page(seed, h, w, rgb=False):
    """A formula-like page: white with dark random strokes (uint8)."""
    a = np.full((h, w, 3 if rgb else 1), 255, np.int64)
    n = 12 + seed % 7
    ys, xs = synth.randint(seed, "y", (n,), 2, h - 6), synth.randint(seed, "x", (n,), 2, w - 12)
    hh, ww = synth.randint(seed, "h", (n,), 2, 6), synth.randint(seed, "w", (n,), 3, 12)
    ink = synth.randint(seed, "ink", (n, a.shape[2]), 0, 120)
    for y, x, dh, dw, c in zip(ys, xs, hh, ww, ink):
        a[y:y + dh, x:x + dw] = c
    noise = synth.randint(seed, "noise", a.shape, 0, 9)
    a = np.clip(a - noise, 0, 255).astype(np.uint8)
    return a if rgb else a[..., 0]


def inputs():
    """name -> the object handed to Predictor.predict (besides the PNG path)."""
    gray = page(3, 40, 300)
    rgb = page(4, 90, 700, rgb=True)
    wide = page(5, 32, 620)                                        # 64 * 620 / 32 = 1240 > 800: centre crop in load_image
    sized = torch.from_numpy(page(6, 64, 800).astype(np.float32) / 255.0)          # (64, 800) in [0, 1]: used as is
    return {
        "pil_gray": Image.fromarray(gray, "L"), "pil_rgb": Image.fromarray(rgb, "RGB"),
        "pil_sized": Image.fromarray(page(7, 64, 800), "L"),
        "np_u8_hw": gray, "np_u8_hwc": rgb[..., :1].copy(), "np_f32_chw": (wide[None].astype(np.float32)),
        "tensor_sized_01": sized, "tensor_2d_255": torch.from_numpy(gray.astype(np.float32)),
        "tensor_chw_pm1": torch.from_numpy(wide[None].astype(np.float32) / 127.5 - 1.0),
    }
