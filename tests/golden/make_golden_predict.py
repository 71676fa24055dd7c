#!/usr/bin/env python3
"""Write tests/golden/predict_64x800.pt + predict_page.png + predict_64x800.npz: the `predict CHECKPOINT IMAGE` chain
of the reference (cli.py:253-308 -> Predictor.from_checkpoint -> Predictor.predict, predictor.py:61-203 ->
_prepare_image, :396-462 -> load_image, data/utils.py:18-90) for a checkpoint whose model takes the 1x64x800 input
`_prepare_image` hard-codes (:409-414).

Runs only in the build container (needs /root/reference; imported unmodified behind the same inert `torchvision`
shim as make_golden.py).  The checkpoint FILE is written by the reference's own Trainer.save_checkpoint (as in
make_golden_checkpoint.py) from repo-generated weights; the page image is a synthetic PNG (random dark strokes on
white).  Expected values = what the reference's Predictor does with them, for every input type `_prepare_image`
accepts: a path, a PIL image, uint8 / float numpy arrays, sized and un-sized tensors.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_predict.py
"""
import json
import os
import shutil
import sys
import tempfile
import types
from pathlib import Path

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(REPO, "hmer-img2latex_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
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
import torch  # noqa: E402
from PIL import Image  # noqa: E402

from img2latex_amd import synth  # noqa: E402
from predict_inputs import inputs, page  # noqa: E402  (tests/golden/predict_inputs.py: shared with the tests)

logging.disable(logging.CRITICAL)
from img2latex.data.tokenizer import LaTeXTokenizer  # noqa: E402  (the reference)
from img2latex.model import Seq2SeqModel  # noqa: E402
from img2latex.training import trainer as trainer_mod  # noqa: E402
from img2latex.training.predictor import Predictor  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CFG_KW = dict(vocab_size=50, embedding_dim=32, hidden_dim=64, lstm_layers=1, attention=False, channels=1,
              img_height=64, img_width=800, conv_filters=(2, 4, 8))
SD_KW = dict(seed=21, out_scale=12.0, enc_scale=24.0, end_clock=(0.05, 12.0, 6.0))


def main():
    cfg = synth.model_config(**CFG_KW)
    tok = LaTeXTokenizer(max_sequence_length=150)
    tok.fit([" ".join([f"t{i}"] * (100 - i)) for i in range(4, cfg["vocab_size"])])
    assert tok.vocab_size == cfg["vocab_size"]
    model = Seq2SeqModel(model_type="cnn_lstm", vocab_size=tok.vocab_size, encoder_params=synth.encoder_params(cfg),
                         decoder_params=synth.decoder_params(cfg))
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, **SD_KW).items()})
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)              # no step taken: empty state
    config = {"model": {"name": "cnn_lstm", "embedding_dim": cfg["embedding_dim"],
                        "encoder": {"cnn": {k: (list(v) if isinstance(v, tuple) else v)
                                            for k, v in synth.encoder_params(cfg).items() if k != "embedding_dim"}},
                        "decoder": synth.decoder_params(cfg)},
              "training": {"learning_rate": 1e-3, "weight_decay": 1e-4, "clip_grad_norm": 5.0}}
    tmp = Path(tempfile.mkdtemp(prefix="i2l_pred_"))
    try:
        trainer_mod.experiment_registry = types.SimpleNamespace(
            path_manager=types.SimpleNamespace(get_checkpoint_dir=lambda name: tmp))
        tr = trainer_mod.Trainer.__new__(trainer_mod.Trainer)
        tr.model, tr.optimizer, tr.tokenizer, tr.config, tr.experiment_name = model, opt, tok, config, "fixture"
        path = tr.save_checkpoint(epoch=0, step=0, metrics={}, is_best=False)
        dst = os.path.join(OUT, "predict_64x800.pt")
        shutil.copyfile(path, dst)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    png = os.path.join(OUT, "predict_page.png")
    Image.fromarray(page(1, 47, 410), "L").save(png, optimize=True)
    png_rgb = os.path.join(OUT, "predict_page_rgb.png")
    Image.fromarray(page(2, 30, 500, rgb=True), "RGB").save(png_rgb, optimize=True)

    pred = Predictor.from_checkpoint(dst, device=torch.device("cpu"))
    out = {"cfg_json": np.array(json.dumps(cfg))}
    cases = {"path_gray_png": png, "path_rgb_png": png_rgb, "path_missing": os.path.join(OUT, "no_such_file.png")}
    cases.update(inputs())
    texts, names = {}, []
    for name, obj in cases.items():
        prepared = pred._prepare_image(obj)                          # predictor.py:396-462
        assert tuple(prepared.shape) == (1, 1, 64, 800), (name, prepared.shape)
        flat = prepared.reshape(-1)
        out[f"prep_{name}_sample"] = flat[::97].numpy().copy()
        out[f"prep_{name}_checksum"] = np.array(synth.checksum(flat.numpy()))
        texts[name] = pred.predict(obj, max_length=40)               # predictor.py:139-203
        texts[name + "@T0.7"] = pred.predict(obj, max_length=25, temperature=0.7)
        names.append(name)
    order = ["path_gray_png", "pil_rgb", "np_u8_hw", "tensor_sized_01", "path_rgb_png"]
    texts["predict_batch"] = pred.predict_batch([cases[n] for n in order], max_length=40, batch_size=3)
    out["batch_order"] = np.array(json.dumps(order))
    out["names"] = np.array(json.dumps(names))
    out["texts"] = np.array(json.dumps(texts))
    np.savez_compressed(os.path.join(OUT, "predict_64x800.npz"), **out)
    print("predict_64x800.pt", os.path.getsize(dst), "bytes;", os.path.getsize(png), "+", os.path.getsize(png_rgb), "bytes of PNG")
    for k, v in texts.items():
        print(f"  {k:22s} {v if isinstance(v, str) else v[:2]}")


if __name__ == "__main__":
    main()
