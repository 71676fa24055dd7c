#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by running the REAL reference.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  The reference package is imported UNMODIFIED from where it lies;
the one shim is an inert ``torchvision`` module, because encoder.py:9 imports it
at module top (and data/transforms.py:5 imports torchvision.transforms.functional)
and torchvision is not installed.  Weights and
inputs come from the repo-owned counter-based generator (img2latex_amd.synth),
so the fixtures hold only expected OUTPUTS plus the generator arguments.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixture families: G1 encoder blocks, G2 decode_step x3,
G3 greedy ids (+margins), G4 beam ids+scores, G5 Predictor loop ids,
G6 teacher-forced logits / CE loss / grads / one Adam step, G7 attention identity.
"""
import json
import os
import sys
import types

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(REPO, "hmer-img2latex_amd"))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

# inert torchvision: four empty modules, nothing in them is ever called on this path
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
from img2latex.model import Seq2SeqModel  # noqa: E402  (the reference)
from img2latex.training.predictor import Predictor  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
START, END, PAD = synth.START, synth.END, synth.PAD

CONFIGS = {
    # name: (cfg kwargs, state-dict kwargs)
    "tiny_l1": (dict(vocab_size=50, embedding_dim=32, hidden_dim=64, lstm_layers=1, attention=False,
                     channels=1, img_height=16, img_width=32, conv_filters=(4, 8, 16)),
                dict(seed=7, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))),
    "tiny_l2_attn": (dict(vocab_size=50, embedding_dim=32, hidden_dim=64, lstm_layers=2, attention=True,
                          channels=1, img_height=16, img_width=32, conv_filters=(4, 8, 16)),
                     dict(seed=8, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))),
    "odd_dims": (dict(vocab_size=37, embedding_dim=32, hidden_dim=64, lstm_layers=1, attention=True,
                      channels=3, img_height=22, img_width=50, conv_filters=(5, 7, 9)),
                 dict(seed=9, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))),
    "primary": (dict(), dict(seed=42, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))),
    "secondary": (dict(embedding_dim=512, hidden_dim=512, lstm_layers=2, attention=True),
                  dict(seed=43, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))),
    # the reference's OWN shapes (r03): its CNN test vector (4,1,64,800) -> (4,256) (tests/test_encoder.py:11-42,
    # constructor defaults for everything else) ...
    "ref_test_64x800": (dict(channels=1, img_height=64, img_width=800),
                        dict(seed=44, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))),
    # ... and the shipped cnn_lstm configuration (configs/config.yaml:30-50): 1x128x800, E = Hd = 512, L = 2, attention;
    # Flatten feeds the Linear K = 128 * 16 * 100 = 204 800 features
    "shipped_128x800": (dict(channels=1, img_height=128, img_width=800, embedding_dim=512, hidden_dim=512,
                             lstm_layers=2, attention=True),
                        dict(seed=45, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))),
}
BIG = ("primary", "secondary", "ref_test_64x800", "shipped_128x800")


def build_reference(cfg, np_sd):
    m = Seq2SeqModel(model_type="cnn_lstm", vocab_size=cfg["vocab_size"],
                     encoder_params=synth.encoder_params(cfg),
                     decoder_params=synth.decoder_params(cfg))
    missing = m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in np_sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    m.eval()
    return m


class IdTokenizer:
    """Stands in for LaTeXTokenizer in Predictor: only the three attributes the
    greedy loop touches (predictor.py:248-249,391); decode() returns the ids as JSON."""
    start_token_id, end_token_id, pad_token_id = START, END, PAD

    def decode(self, ids):
        return json.dumps([int(i) for i in ids])


def sample(t, n=4096):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy()


def run_config(name, cfg_kw, sd_kw, big):
    cfg = synth.model_config(**cfg_kw)
    np_sd = synth.make_state_dict(cfg, **sd_kw)
    ref = build_reference(cfg, np_sd)
    out = {"cfg_json": np.array(json.dumps(cfg)), "sd_kw_json": np.array(json.dumps(sd_kw))}
    B = 4
    imgs = torch.from_numpy(synth.make_images(B, cfg, seed=1234))
    with torch.no_grad():
        # ---- G1 encoder blocks (encoder.py:111-129)
        x = imgs
        blk = 0
        for i, layer in enumerate(ref.encoder.cnn_layers):
            x = layer(x)
            if i % 3 == 2:
                out[f"g1_block{blk}_shape"] = np.array(x.shape)
                out[f"g1_block{blk}_checksum"] = np.array(synth.checksum(x.numpy()))
                out[f"g1_block{blk}_sample" if big else f"g1_block{blk}"] = sample(x) if big else x.numpy().copy()
                blk += 1
        enc = ref.encoder(imgs)
        out["g1_enc"] = enc.numpy().copy()

        # ---- G2 three consecutive decode_step calls (decoder.py:197-284)
        tok = torch.full((B, 1), START, dtype=torch.long)
        hidden = None
        for s in range(3):
            logits, hidden = ref.decoder.decode_step(enc, tok, hidden)
            out[f"g2_logits{s}"] = logits.numpy().copy()
            out[f"g2_h{s}"] = hidden[0].numpy().copy()
            out[f"g2_c{s}"] = hidden[1].numpy().copy()
            tok = logits.squeeze(1).argmax(-1, keepdim=True)
            out[f"g2_tok{s}"] = tok.numpy().copy()

        # ---- G7 attention identity (decoder.py:312-343 with src_len == 1)
        if cfg["attention"]:
            ctx = ref.decoder.attention(hidden[0][-1].unsqueeze(1), enc.unsqueeze(1))
            out["g7_identity"] = np.array(bool(torch.equal(ctx.squeeze(1), enc)))

        # ---- G3 greedy (seq2seq.py:192-232): cfg1 = B 4, 32 steps
        ids = ref.inference(imgs, START, END, max_length=32)
        out["g3_b4_ids"] = np.array(ids, dtype=np.int64)
        one = ref.inference(imgs[1:2], START, END, max_length=32)      # B == 1 post-processing
        out["g3_b1_ids"] = np.array(one, dtype=np.int64)
        ids_t = ref.inference(imgs, START, END, max_length=12, temperature=0.7)
        out["g3_b4_temp_ids"] = np.array(ids_t, dtype=np.int64)

        # ---- G4 beam search (seq2seq.py:234-298), B == 1, k = 5 and k = 3
        nb = 8
        bimgs = torch.from_numpy(synth.make_images(nb, cfg, seed=4321))
        for k in (5, 3):
            seqs, lens = [], []
            for j in range(nb):
                s = ref.inference(bimgs[j:j + 1], START, END, max_length=40, beam_size=k)
                seqs.append(s)
                lens.append(len(s))
            arr = np.full((nb, max(lens + [1])), -1, dtype=np.int64)
            for j, s in enumerate(seqs):
                arr[j, :len(s)] = s
            out[f"g4_k{k}_ids"] = arr
            out[f"g4_k{k}_len"] = np.array(lens)
        # beam on a batch > 1 falls back to greedy (seq2seq.py:244-247)
        out["g4_fallback_ids"] = np.array(ref.inference(imgs, START, END, max_length=10, beam_size=3), dtype=np.int64)

        # ---- G5 Predictor.predict_batch greedy loop (predictor.py:254-361)
        pred = Predictor(ref, IdTokenizer(), device=torch.device("cpu"), model_type="cnn_lstm")
        pred._prepare_image = lambda im: im.unsqueeze(0)       # identity image prep (tensor already sized)
        res = pred.predict_batch([imgs[i] for i in range(B)], max_length=32, batch_size=B)
        seqs = [json.loads(r) for r in res]
        arr = np.full((B, max(len(s) for s in seqs) + 1), -1, dtype=np.int64)
        for j, s in enumerate(seqs):
            arr[j, :len(s)] = s
        out["g5_ids"] = arr                                       # START stripped at predictor.py:384-385
        out["g5_len"] = np.array([len(s) for s in seqs])

    # ---- G6 teacher forcing + CE + one optimiser step (trainer.py:91-115,334-343), dropout 0
    T = 12 if not big else 24
    forms = torch.from_numpy(synth.make_formulas(B, T, cfg["vocab_size"], seed=777, min_len=5))
    ref.train()                                                   # dropout p = 0 in these configs
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3, weight_decay=1e-4)
    crit = torch.nn.CrossEntropyLoss(ignore_index=PAD, reduction="mean", label_smoothing=0.1)
    opt.zero_grad(set_to_none=True)
    outputs = ref(imgs, forms)
    loss = crit(outputs.transpose(1, 2), forms[:, 1:])
    loss.backward()
    names = [n for n, _ in ref.named_parameters()]
    out["g6_logits" if not big else "g6_logits_sample"] = (outputs.detach().numpy().copy() if not big else sample(outputs))
    out["g6_logits_checksum"] = np.array(synth.checksum(outputs.detach().numpy()))
    out["g6_loss"] = np.array(float(loss))
    out["g6_param_names"] = np.array(json.dumps(names))
    out["g6_grad_norms"] = np.array([float(p.grad.norm()) for _, p in ref.named_parameters()])
    out["g6_grad_checksums"] = np.array([synth.checksum(p.grad.numpy()) for _, p in ref.named_parameters()])
    total = torch.nn.utils.clip_grad_norm_(ref.parameters(), 5.0)
    out["g6_total_norm"] = np.array(float(total))
    opt.step()
    out["g6_param_checksums_after"] = np.array([synth.checksum(p.detach().numpy()) for _, p in ref.named_parameters()])
    out["g6_param_sample_after"] = np.stack([sample(p, 8)[:8] if p.numel() >= 8 else np.resize(p.detach().numpy().ravel(), 8)
                                             for _, p in ref.named_parameters()])
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(name, "ok", {k: (v.shape if hasattr(v, "shape") else None) for k, v in list(out.items())[:3]})
    return ref, cfg, np_sd


CFG2_SD_KW = {
    # END never wins -> all 150 steps run (the bench workload, BASELINE config 2)
    "primary_cfg2": dict(seed=42, out_scale=8.0, enc_scale=16.0),
    # with the END clock: exercises the "all rows emit END in the same step" stop (seq2seq.py:220)
    "primary_cfg2_clock": dict(seed=42, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0)),
}


def run_cfg2(fname, cfg, sd_kw):
    """BASELINE config 2 shapes: B=256, up to 150 greedy steps, ids + top1-top2 margins."""
    B, T = 256, 150
    imgs = torch.from_numpy(synth.make_images(B, cfg, seed=1234))
    ref = build_reference(cfg, synth.make_state_dict(cfg, **sd_kw))
    with torch.no_grad():
        enc = ref.encoder(imgs)
        ids = ref._greedy_search(enc, START, END, T, 1.0, 0, 0.0)
        # margins: replay the reference decode_step along the reference's own ids
        tok = torch.full((B, 1), START, dtype=torch.long)
        hidden = None
        arr = np.array(ids, dtype=np.int64)
        margins = np.zeros((B, arr.shape[1] - 1), dtype=np.float32)
        for t in range(arr.shape[1] - 1):
            logits, hidden = ref.decoder.decode_step(enc, tok, hidden)
            top2 = torch.topk(logits.squeeze(1), 2, dim=-1).values
            margins[:, t] = (top2[:, 0] - top2[:, 1]).numpy()
            tok = torch.from_numpy(arr[:, t + 1:t + 2].copy())
    np.savez_compressed(os.path.join(OUT, fname + ".npz"), ids=arr.astype(np.int16), margins=margins,
                        sd_kw_json=np.array(json.dumps(sd_kw)), cfg_json=np.array(json.dumps(cfg)),
                        enc_checksum=np.array(synth.checksum(enc.numpy())), enc_sample=sample(enc, 1024))
    print(fname, "ok", arr.shape, "min margin", margins.min(), "END count", int((arr == END).sum()))


def traced_beam_search(ref, image, k, max_length):
    """Run the reference's UNMODIFIED Seq2SeqModel.inference(beam_size=k) on one image and read, through a trace
    function on the `_beam_search` frame (seq2seq.py:234-298), what the method does not return: the winning beam's
    score (:286-290) and, per step, the smallest gap between neighbours among the k + 1 best candidate scores right
    after the sort (:279-280) -- the decisions an fp32 reassociation of the logits could flip."""
    import inspect
    fn = Seq2SeqModel._beam_search
    code = fn.__code__
    src, first = inspect.getsourcelines(fn)
    cut_line = first + next(i for i, ln in enumerate(src) if "beams = candidates[:beam_size]" in ln)
    info = {"score": None, "gap": float("inf"), "final_gap": float("inf"), "steps": 0}

    def local(frame, event, arg):
        if event == "line" and frame.f_lineno == cut_line:
            sc = [c["score"] for c in frame.f_locals["candidates"][:k + 1]]
            info["steps"] += 1
            for a, b in zip(sc[:-1], sc[1:]):
                info["gap"] = min(info["gap"], a - b)
        elif event == "return":
            best = frame.f_locals.get("best_beam")
            if best is not None:
                info["score"] = float(best["score"])
                done = sorted((c["score"] for c in frame.f_locals["completed"]), reverse=True)
                if len(done) > 1:
                    info["final_gap"] = done[0] - done[1]
        return local

    def tracer(frame, event, arg):
        return local if frame.f_code is code else None

    sys.settrace(tracer)
    try:
        seq = ref.inference(image, START, END, max_length=max_length, beam_size=k)
    finally:
        sys.settrace(None)
    return seq, info


def run_cfg3_beam(fname, cfg, sd_kw, n, k, T, seed=1234):
    """BASELINE configs[2] at ITS size: n single-image beam searches (k beams, up to T steps, attention on) by the
    reference, tokens + winning score + decision margins per image."""
    imgs = torch.from_numpy(synth.make_images(n, cfg, seed=seed))
    ref = build_reference(cfg, synth.make_state_dict(cfg, **sd_kw))
    seqs, scores, gaps, fgaps, steps = [], [], [], [], []
    with torch.no_grad():
        for j in range(n):
            s, info = traced_beam_search(ref, imgs[j:j + 1], k, T)
            seqs.append(s)
            scores.append(info["score"])
            gaps.append(info["gap"])
            fgaps.append(info["final_gap"])
            steps.append(info["steps"])
        enc = ref.encoder(imgs)
    arr = np.full((n, max(len(s) for s in seqs) + 1), -1, dtype=np.int16)
    for j, s in enumerate(seqs):
        arr[j, :len(s)] = s
    np.savez_compressed(os.path.join(OUT, fname + ".npz"), ids=arr, lens=np.array([len(s) for s in seqs]),
                        scores=np.array(scores, dtype=np.float64), min_gap=np.array(gaps), final_gap=np.array(fgaps),
                        steps=np.array(steps), k=np.array(k), max_length=np.array(T), image_seed=np.array(seed),
                        sd_kw_json=np.array(json.dumps(sd_kw)), cfg_json=np.array(json.dumps(cfg)),
                        enc_checksum=np.array(synth.checksum(enc.numpy())))
    print(fname, "ok", arr.shape, "steps", min(steps), "..", max(steps), "min gap", min(gaps), "min final gap", min(fgaps))


CFG3_BEAM = {
    # = bench.py --mode beam: 128 images, k = 5, max_length 150, attention, END clock (searches end at row-dependent steps)
    "primary_cfg3_beam": (dict(seed=42, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0)), 128, 5, 150),
    # END never wins: every search runs all 150 steps with 5 live beams (completed stays empty -> beams[0])
    "primary_cfg3_beam_noend": (dict(seed=42, out_scale=8.0, enc_scale=16.0), 32, 5, 150),
}


def run_init_parity():
    """Same torch.manual_seed -> same default init as the reference constructors
    (parameter creation order encoder.py:78-106, decoder.py:69-90)."""
    cfg = synth.model_config(vocab_size=50, embedding_dim=32, hidden_dim=64, lstm_layers=2, attention=True,
                             channels=1, img_height=16, img_width=32, conv_filters=(4, 8, 16))
    torch.manual_seed(1234)
    m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
    names = list(m.state_dict().keys())
    np.savez_compressed(os.path.join(OUT, "init_parity.npz"), cfg_json=np.array(json.dumps(cfg)),
                        names=np.array(json.dumps(names)),
                        checksums=np.array([synth.checksum(v.numpy()) for v in m.state_dict().values()]))
    print("init parity ok", len(names))


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for name, (ckw, skw) in CONFIGS.items():
        if only and name not in only:
            continue
        ref, cfg, _ = run_config(name, ckw, skw, big=name in BIG)
        if name == "primary":
            for fname, kw in CFG2_SD_KW.items():
                run_cfg2(fname, cfg, kw)
    for fname, (kw, n, k, T) in CFG3_BEAM.items():
        if only and fname not in only:
            continue
        run_cfg3_beam(fname, synth.model_config(attention=True), kw, n, k, T)
    if not only:
        run_init_parity()
