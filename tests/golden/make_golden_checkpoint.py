#!/usr/bin/env python3
"""Write tests/golden/ref_checkpoint.pt + ref_checkpoint.npz: a checkpoint file produced by the REFERENCE's own
``Trainer.save_checkpoint`` (trainer.py:181-233) with the reference's own ``LaTeXTokenizer`` and a real
``torch.optim.Adam`` state, plus what the reference then does with it.

Runs only in the build container (needs /root/reference).  The reference is imported unmodified (same inert
``torchvision`` shim as make_golden.py).  ``Trainer.__init__`` needs data loaders, the experiment registry and a
device probe, none of which is on this path, so the Trainer object is created without it and given exactly the
attributes ``save_checkpoint`` reads (model, optimizer, tokenizer, config, experiment_name); the module's
``experiment_registry`` name is pointed at a stand-in whose ``path_manager.get_checkpoint_dir`` returns a temporary
directory, because the real one would create ``outputs/<experiment>/checkpoints`` under /root/reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_checkpoint.py

Contents (tiny_l1 dims so that the file stays ~0.5 MB):
  ref_checkpoint.pt   the dict of trainer.py:209-224 after ONE reference optimisation step (fixture G6's step):
                      model_state_dict, optimizer_state_dict (Adam step 1), config, tokenizer_config
  ref_checkpoint.npz  expected behaviour of the reference AFTER loading that file:
                      ck_ids / ck_len   Predictor.from_checkpoint(file).predict_batch ids (predictor.py:205-394)
                      ck_text           the decoded strings (LaTeXTokenizer.decode)
                      ck_resume_*       Trainer.load_checkpoint semantics (:235-269): a SECOND optimisation step from
                                        the restored model + optimizer state: loss and parameter samples after it
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

from img2latex_amd import synth  # noqa: E402

logging.disable(logging.CRITICAL)
from img2latex.data.tokenizer import LaTeXTokenizer  # noqa: E402  (the reference)
from img2latex.model import Seq2SeqModel  # noqa: E402
from img2latex.training import trainer as trainer_mod  # noqa: E402
from img2latex.training.predictor import Predictor  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CFG_KW = dict(vocab_size=50, embedding_dim=32, hidden_dim=64, lstm_layers=1, attention=False, channels=1,
              img_height=16, img_width=32, conv_filters=(4, 8, 16))
SD_KW = dict(seed=7, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))


def sample(t, n=8):
    f = t.detach().reshape(-1)
    if f.numel() < n:
        return np.resize(f.numpy(), n)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def one_step(model, opt, imgs, forms):
    crit = torch.nn.CrossEntropyLoss(ignore_index=synth.PAD, reduction="mean", label_smoothing=0.1)
    opt.zero_grad(set_to_none=True)
    loss = crit(model(imgs, forms).transpose(1, 2), forms[:, 1:])
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    opt.step()
    return float(loss)


def main():
    cfg = synth.model_config(**CFG_KW)
    # the reference's tokenizer, fitted so that its frequency-sorted vocabulary is 4 specials + t4..t49 in id order
    tok = LaTeXTokenizer(max_sequence_length=150)
    tok.fit([" ".join([f"t{i}"] * (100 - i)) for i in range(4, cfg["vocab_size"])])
    assert tok.vocab_size == cfg["vocab_size"] and tok.token_to_id["t4"] == 4 and tok.token_to_id["t49"] == 49
    model = Seq2SeqModel(model_type="cnn_lstm", vocab_size=tok.vocab_size, encoder_params=synth.encoder_params(cfg),
                         decoder_params=synth.decoder_params(cfg))
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.make_state_dict(cfg, **SD_KW).items()})
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    imgs = torch.from_numpy(synth.make_images(4, cfg, seed=1234))
    forms = torch.from_numpy(synth.make_formulas(4, 12, cfg["vocab_size"], seed=777, min_len=5))
    one_step(model, opt, imgs, forms)

    # config in the layout cli.train passes to the Trainer and Predictor.from_checkpoint reads back (predictor.py:83-126)
    config = {"model": {"name": "cnn_lstm", "embedding_dim": cfg["embedding_dim"],
                        "encoder": {"cnn": {k: (list(v) if isinstance(v, tuple) else v)
                                            for k, v in synth.encoder_params(cfg).items() if k != "embedding_dim"}},
                        "decoder": synth.decoder_params(cfg)},
              "training": {"learning_rate": 1e-3, "weight_decay": 1e-4, "clip_grad_norm": 5.0}}
    tmp = Path(tempfile.mkdtemp(prefix="i2l_ck_"))
    try:
        trainer_mod.experiment_registry = types.SimpleNamespace(
            path_manager=types.SimpleNamespace(get_checkpoint_dir=lambda name: tmp))
        tr = trainer_mod.Trainer.__new__(trainer_mod.Trainer)
        tr.model, tr.optimizer, tr.tokenizer, tr.config, tr.experiment_name = model, opt, tok, config, "fixture"
        path = tr.save_checkpoint(epoch=1, step=1, metrics={"loss": 0.0}, is_best=False)     # the reference writes it
        dst = os.path.join(OUT, "ref_checkpoint.pt")
        shutil.copyfile(path, dst)

        # what the reference does with the file: Predictor.from_checkpoint -> predict_batch
        pred = Predictor.from_checkpoint(dst, device=torch.device("cpu"))
        pred._prepare_image = lambda im: im.unsqueeze(0)            # tensors are already sized for the model
        ids_tok = types.SimpleNamespace(start_token_id=tok.start_token_id, end_token_id=tok.end_token_id,
                                        pad_token_id=tok.pad_token_id, decode=lambda ids: json.dumps([int(i) for i in ids]))
        real_tok, pred.tokenizer = pred.tokenizer, ids_tok
        seqs = [json.loads(r) for r in pred.predict_batch([imgs[i] for i in range(4)], max_length=32, batch_size=4)]
        pred.tokenizer = real_tok
        texts = pred.predict_batch([imgs[i] for i in range(4)], max_length=32, batch_size=4)
        arr = np.full((4, max(len(s) for s in seqs) + 1), -1, dtype=np.int64)
        for j, s in enumerate(seqs):
            arr[j, :len(s)] = s

        # resume (Trainer.load_checkpoint, trainer.py:235-269): fresh model + optimizer from the file, second step
        ck = torch.load(dst, map_location="cpu", weights_only=False)
        m2 = Seq2SeqModel(model_type="cnn_lstm", vocab_size=tok.vocab_size, encoder_params=synth.encoder_params(cfg),
                          decoder_params=synth.decoder_params(cfg))
        m2.load_state_dict(ck["model_state_dict"])
        m2.train()
        o2 = torch.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=1e-4)
        o2.load_state_dict(ck["optimizer_state_dict"])
        forms2 = torch.from_numpy(synth.make_formulas(4, 12, cfg["vocab_size"], seed=778, min_len=5))
        loss2 = one_step(m2, o2, imgs, forms2)
        names = [n for n, _ in m2.named_parameters()]
        np.savez_compressed(os.path.join(OUT, "ref_checkpoint.npz"), cfg_json=np.array(json.dumps(cfg)),
                            ck_ids=arr, ck_len=np.array([len(s) for s in seqs]), ck_text=np.array(json.dumps(texts)),
                            ck_resume_loss=np.array(loss2), ck_param_names=np.array(json.dumps(names)),
                            ck_resume_param_sample=np.stack([sample(p) for _, p in m2.named_parameters()]))
        print("ref_checkpoint.pt", os.path.getsize(dst), "bytes; keys", sorted(ck), "; texts", texts[:2])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
