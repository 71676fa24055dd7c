"""The two entry points of the reference's CLI that sit on the hot path, with its argument and option names
(img2latex/cli.py:104-120 ``train``, :253-269 ``predict``):

    python -m img2latex_amd predict CHECKPOINT IMAGE [--beam-size N] [--max-length N] [--temperature T]
                                                     [--top-k K] [--top-p P] [--device cuda]
    python -m img2latex_amd train [--config-path F] [--experiment-name S] [--checkpoint-path F] [--data-dir D]
                                  [--device cuda] [--seed N]

``predict`` is the reference's chain end to end on the device: Predictor.from_checkpoint (predictor.py:61-137) ->
Predictor.predict(image_path) (:139-203) -> load_image (data/utils.py:18-90) -> encoder -> greedy search -> string.

``train`` builds the model from the YAML exactly as cli.py:195-217 does, resumes model + Adam state from a reference
checkpoint (trainer.py:235-269) and drives ``TrainStep`` (the optimisation step of trainer.py:303-343 on the HIP
kernels) over the batches of the data loaders.  Dataset classes, file decoding and vocabulary fitting are outside this
package's scope: the batches come from the reference's own ``create_data_loaders`` /
``LaTeXTokenizer`` when the ``img2latex`` package is importable beside this one, or -- ``--synthetic-steps N`` -- from
the deterministic generator the benchmarks use.  Experiment registry, logging setup, rich console output, validation
metrics and early stopping are the reference's host-side orchestration and are not rebuilt here.

The device is a ROCm GPU; ``--device cpu`` is refused (there is no CPU fallback in this package).
"""
from __future__ import annotations

import argparse
import os
import sys
from typing import Dict, List, Optional

import torch


def _device(name: Optional[str]) -> torch.device:
    """cli.py:266-268 / mps_utils.set_device, minus the silent fall-back to the CPU."""
    if name not in (None, "cuda") and not str(name).startswith("cuda:"):
        raise SystemExit(f"img2latex_amd: --device {name}: this package runs on a ROCm GPU only (no CPU / MPS path)")
    if not torch.cuda.is_available():
        raise SystemExit("img2latex_amd: no ROCm device is visible")
    return torch.device(name or "cuda")


def predict(checkpoint_path: str, image_path: str, beam_size: int = 0, max_length: int = 141, temperature: float = 1.0,
            top_k: int = 0, top_p: float = 0.0, device: Optional[str] = None) -> str:
    """cli.py:253-308: returns the LaTeX string the command prints."""
    from .training import Predictor
    predictor = Predictor.from_checkpoint(checkpoint_path=checkpoint_path, device=_device(device))
    return predictor.predict(image=image_path, beam_size=beam_size, max_length=max_length, temperature=temperature,
                             top_k=top_k, top_p=top_p)


def load_config(config_path: str) -> Dict:
    """cli.py:83-101."""
    import yaml
    if not os.path.exists(config_path):
        raise FileNotFoundError(f"Config file not found: {config_path}")
    with open(config_path, "r") as f:
        return yaml.safe_load(f)


def build_model(config: Dict, vocab_size: int):
    """cli.py:195-217: config["model"] -> Seq2SeqModel."""
    from .model import Seq2SeqModel
    model_cfg = config["model"]
    enc = dict(model_cfg["encoder"]["cnn" if model_cfg["name"] == "cnn_lstm" else "resnet"])
    enc["embedding_dim"] = model_cfg["embedding_dim"]
    return Seq2SeqModel(model_type=model_cfg["name"], vocab_size=vocab_size, encoder_params=enc,
                        decoder_params=model_cfg["decoder"])


def _synthetic_batches(config: Dict, steps: int, vocab_size: int, seed: int):
    from . import synth
    model_cfg, batch = config["model"], int(config.get("data", {}).get("batch_size", 64))
    enc = model_cfg["encoder"]["cnn" if model_cfg["name"] == "cnn_lstm" else "resnet"]
    length = int(config.get("data", {}).get("max_seq_length", 150))
    for s in range(steps):
        images = synth.uniform(seed + s, "images", (batch, enc["channels"], enc["img_height"], enc["img_width"]), -1.0, 1.0)
        yield {"images": torch.from_numpy(images),
               "formulas": torch.from_numpy(synth.make_formulas(batch, length, vocab_size, seed=seed + s))}


def train(config_path: str = "img2latex/configs/config.yaml", experiment_name: str = "img2latex_v1",
          checkpoint_path: Optional[str] = None, data_dir: Optional[str] = None, device: Optional[str] = None,
          seed: int = 42, synthetic_steps: int = 0, synthetic_vocab: int = 512, output_dir: str = "outputs") -> Dict:
    """cli.py:104-250 reduced to the hot path: model from the config, optional resume, TrainStep over the batches, a
    checkpoint in the reference's layout (trainer.py:209-224) per epoch.  Returns {"loss", "steps", "checkpoint"}."""
    from . import data as D
    from .training import TokenTable, TrainStep, save_checkpoint
    torch.manual_seed(seed)                                                 # mps_utils.set_seed
    config = load_config(config_path)
    if data_dir:
        config.setdefault("data", {})["data_dir"] = data_dir
    config.setdefault("training", {})["experiment_name"] = experiment_name
    dev = _device(device or config["training"].get("device"))
    tcfg = config["training"]
    resume = torch.load(checkpoint_path, map_location="cpu", weights_only=False) if checkpoint_path else None
    if synthetic_steps > 0:
        if resume is not None:
            tk = resume["tokenizer_config"]
            tokenizer = TokenTable(tk.get("token_to_id"), tk.get("special_tokens"), tk.get("max_sequence_length", 141))
        else:
            vocab = {"<PAD>": 0, "<START>": 1, "<END>": 2, "<UNK>": 3}
            vocab.update({f"t{i}": i for i in range(4, synthetic_vocab)})
            tokenizer = TokenTable(vocab, max_sequence_length=int(config.get("data", {}).get("max_seq_length", 150)))
        max_epochs, loaders = None, None                                    # one pass over the synthetic batches
    else:
        try:
            from img2latex.data.dataset import create_data_loaders          # the reference's own dataset + tokenizer
            from img2latex.data.tokenizer import LaTeXTokenizer
        except ImportError as exc:
            raise SystemExit("img2latex_amd train: datasets and vocabulary fitting are the reference's (img2latex.data."
                             "dataset / tokenizer, outside this package's scope); install it beside this package, or "
                             f"pass --synthetic-steps N.  ({exc})")
        tokenizer = LaTeXTokenizer(max_sequence_length=config["data"]["max_seq_length"])
        tokenizer.fit_on_formulas_file(os.path.join(config["data"]["data_dir"], config["data"]["formulas_file"]))
        loaders = create_data_loaders(config=config, tokenizer=tokenizer, max_samples=None)
        max_epochs = int(tcfg.get("epochs", 50))                            # trainer.py:118
    model = build_model(config, tokenizer.vocab_size)
    if resume is not None:
        model.load_state_dict(resume["model_state_dict"])                   # trainer.py:253
    model = model.to(dev)
    ts = TrainStep(model, lr=float(tcfg.get("learning_rate", 1e-3)), weight_decay=float(tcfg.get("weight_decay", 1e-4)),
                   clip_grad_norm=float(tcfg.get("clip_grad_norm", 5.0)), pad_token_id=tokenizer.pad_token_id,
                   label_smoothing=0.1, seed=seed)                          # trainer.py:91-93,111-115
    if resume is not None and resume.get("optimizer_state_dict"):
        ts.load_optimizer_state_dict(resume["optimizer_state_dict"])        # trainer.py:255
    # trainer.py:257-258,697: a checkpoint's "epoch" is the number of COMPLETED epochs and "step" the global step; a resumed
    # run continues both and stops at the configured total (it used to restart the step count and always run
    # `epochs` more epochs)
    start_epoch = int(resume.get("epoch", 0)) if resume is not None else 0
    step0 = step = int(resume.get("step", 0)) if resume is not None else 0
    if max_epochs is None:
        max_epochs = start_epoch + 1
    ck_dir = os.path.join(output_dir, experiment_name, "checkpoints")
    os.makedirs(ck_dir, exist_ok=True)
    last, path = None, None
    is_resnet = config["model"]["name"] == "resnet_lstm"
    accum = max(1, int(tcfg.get("accumulation_steps", 1)))                  # trainer.py:86-88 (the shipped config: 4)
    for epoch in range(start_epoch + 1, max_epochs + 1):                    # `epoch` = trainer.py's current_epoch + 1
        batches = _synthetic_batches(config, synthetic_steps, tokenizer.vocab_size, seed) if loaders is None else loaders["train"]
        it = iter(batches)
        batch, batch_idx = next(it, None), 0
        while batch is not None:
            nxt = next(it, None)                                            # one batch of look-ahead: "is this the last one?" (:374)
            images = batch["images"].to(dev)                                # data/utils.py:113-135 prepare_batch
            if is_resnet and images.shape[1] == 1:
                images = D.batch_convert_for_resnet(images)
            update = (batch_idx + 1) % accum == 0 or nxt is None
            last = ts.micro_step(images, batch["formulas"].to(dev), accum, update)
            step += 1                                                       # the reference's global_step counts batches (:399)
            batch, batch_idx = nxt, batch_idx + 1
        loss = float(last["loss"]) if last is not None else float("nan")
        path = os.path.join(ck_dir, f"checkpoint_epoch_{epoch}_step_{step}.pt")
        save_checkpoint(path, model, tokenizer, config, epoch=epoch, step=step, metrics={"loss": loss},
                        optimizer_state_dict=ts.optimizer_state_dict())
        print(f"epoch {epoch}: {step} steps, loss {loss:.4f}, checkpoint {path}")
    return {"loss": float(last["loss"]) if last is not None else None, "steps": step - step0, "global_step": step,
            "checkpoint": path}


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="img2latex_amd", description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="command", required=True)
    p = sub.add_parser("predict", help="Predict LaTeX for an image.")
    p.add_argument("checkpoint_path", help="Path to trained model checkpoint")
    p.add_argument("image_path", help="Path to image file")
    p.add_argument("--beam-size", type=int, default=0, help="Beam size for beam search (0 for greedy search)")
    p.add_argument("--max-length", type=int, default=141, help="Maximum length of the generated sequence")
    p.add_argument("--temperature", type=float, default=1.0, help="Temperature for sampling")
    p.add_argument("--top-k", type=int, default=0, help="Top-k sampling parameter")
    p.add_argument("--top-p", type=float, default=0.0, help="Top-p (nucleus) sampling parameter")
    p.add_argument("--device", default=None, help="Device to use for inference (cuda)")
    t = sub.add_parser("train", help="Train the image-to-LaTeX model.")
    t.add_argument("--config-path", default="img2latex/configs/config.yaml", help="Path to configuration file")
    t.add_argument("--experiment-name", default="img2latex_v1", help="Name of the experiment")
    t.add_argument("--checkpoint-path", default=None, help="Path to checkpoint to resume training from")
    t.add_argument("--data-dir", default=None, help="Path to data directory (overrides config)")
    t.add_argument("--device", default=None, help="Device to use for training (cuda)")
    t.add_argument("--seed", type=int, default=42, help="Random seed for reproducibility")
    t.add_argument("--synthetic-steps", type=int, default=0,
                   help="(this package) train N steps on synthetic batches instead of the reference's data loaders")
    t.add_argument("--synthetic-vocab", type=int, default=512, help="(this package) vocabulary size of the synthetic batches")
    t.add_argument("--output-dir", default="outputs", help="(this package) root of <experiment>/checkpoints")
    args = ap.parse_args(argv)
    if args.command == "predict":
        latex = predict(args.checkpoint_path, args.image_path, args.beam_size, args.max_length, args.temperature,
                        args.top_k, args.top_p, args.device)
        print("Generated LaTeX:")
        print(latex)
        return 0
    try:
        train(args.config_path, args.experiment_name, args.checkpoint_path, args.data_dir, args.device, args.seed,
              args.synthetic_steps, args.synthetic_vocab, args.output_dir)
    except SystemExit:
        raise
    except Exception as exc:                                                # cli.py:247-250: failure -> exit code 1
        print(f"Training failed: {exc}", file=sys.stderr)
        return 1
    return 0
