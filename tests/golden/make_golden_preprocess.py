#!/usr/bin/env python3
"""Golden vectors for the image preprocessing: generated images are written as lossless PNG files to
a temporary directory and pushed through the REAL `img2latex.data.utils.load_image` (reference, imported unmodified
from /root/reference, with Pillow doing decode + LANCZOS resize).  tests/golden/preprocess.npz stores the generator
arguments of every image and the reference's float32 output.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_preprocess.py
"""
import os
import sys
import tempfile
import types

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(REPO, "hmer-img2latex_amd"))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
_tv = types.ModuleType("torchvision")
_tv.__path__ = []
for _sub in ("models", "transforms", "transforms.functional"):
    _m = types.ModuleType("torchvision." + _sub)
    _m.__path__ = []
    sys.modules["torchvision." + _sub] = _m
    setattr(sys.modules["torchvision." + _sub.rsplit(".", 1)[0]] if "." in _sub else _tv, _sub.rsplit(".", 1)[-1], _m)
sys.modules["torchvision"] = _tv

import logging  # noqa: E402

import numpy as np  # noqa: E402
import PIL  # noqa: E402
from PIL import Image  # noqa: E402

from img2latex_amd import synth  # noqa: E402

logging.disable(logging.CRITICAL)
from img2latex.data.utils import load_image  # noqa: E402  (the reference)

# (source h, w, source channels, target (H, W), output channels): down- and up-scaling, pad and crop, both conversions,
# unchanged height (vertical pass skipped), unchanged size (no resampling at all)
CASES = [
    (40, 150, 1, (64, 320), 1), (128, 500, 1, (64, 320), 1), (200, 2400, 1, (64, 320), 1), (64, 320, 1, (64, 320), 1),
    (64, 100, 1, (64, 320), 1), (90, 451, 3, (64, 320), 3), (33, 70, 3, (64, 320), 1), (75, 300, 1, (64, 320), 3),
    (50, 200, 1, (50, 200), 1), (31, 517, 1, (64, 800), 1), (300, 120, 3, (64, 320), 3), (17, 23, 1, (32, 64), 1),
]


def make_image(seed, h, w, c):
    """Formula-like content: white page, dark strokes, some noise; uint8."""
    base = synth.uniform(seed, "img", (h, w, c), 0.0, 1.0)
    strokes = (synth.uniform(seed + 1, "mask", (h, w, 1), 0.0, 1.0) < 0.18)
    img = np.where(strokes, base * 90.0, 200.0 + base * 55.0)
    return np.clip(np.round(img), 0, 255).astype(np.uint8).reshape((h, w) if c == 1 else (h, w, 3))


def main():
    out = {"cases": np.array([[h, w, c, th, tw, oc] for (h, w, c, (th, tw), oc) in CASES], np.int32),
           "pillow": np.array(PIL.__version__)}
    with tempfile.TemporaryDirectory() as tmp:
        for i, (h, w, c, size, oc) in enumerate(CASES):
            img = make_image(1000 + 10 * i, h, w, c)
            path = os.path.join(tmp, f"im{i}.png")
            Image.fromarray(img, "L" if c == 1 else "RGB").save(path)
            ref = load_image(path, img_size=size, channels=oc, normalize=True)
            out[f"out{i}"] = ref.numpy().astype(np.float32)
            if i in (0, 5):
                out[f"raw{i}"] = load_image(path, img_size=size, channels=oc, normalize=False).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "preprocess.npz"), **out)
    print("cases", len(CASES), "pillow", PIL.__version__)


if __name__ == "__main__":
    main()
