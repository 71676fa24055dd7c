"""Deterministic synthetic weights / images / token batches.

Counter-based (splitmix64 finaliser over ``seed, stream(name), index``) and
integer-only up to the final scale, so the container that generates the
golden fixtures and the GPU box regenerate bit-identical float32 arrays with
no dependence on torch's RNG or on libm.  Shapes and distributions follow
SURVEY.md section 8(d): PyTorch-default *distributions* (U(+-1/sqrt(fan_in))
for Conv/Linear/LSTM, ~N(0,1) for the embedding table), images U[-1,1).

Nothing here is copied from the reference; the parameter names are the
reference's state_dict keys (encoder.py:95,106; decoder.py:69-90).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np

PAD, START, END, UNK = 0, 1, 2, 3          # tokenizer.py:36-41,68-78

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for ch in name.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix64(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser, vectorised over uint64."""
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def raw_bits(seed: int, name: str, n: int, lane: int = 0) -> np.ndarray:
    """n uint64 words for (seed, name, lane); counter = element index."""
    base = (seed * 0x9E3779B97F4A7C15 + _fnv1a64(name) * 0xD1B54A32D192ED03
            + lane * 0x8CB92BA72F3D8DD7) & 0xFFFFFFFFFFFFFFFF
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(base)
    return _mix64(_mix64(z))


def u24(seed: int, name: str, n: int, lane: int = 0) -> np.ndarray:
    """Integers in [0, 2^24)."""
    return (raw_bits(seed, name, n, lane) >> np.uint64(40)).astype(np.int64)


def uniform(seed: int, name: str, shape: Sequence[int], lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = u24(seed, name, n).astype(np.float64) / float(1 << 24)      # exact
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal_like(seed: int, name: str, shape: Sequence[int]) -> np.ndarray:
    """~N(0,1): Irwin-Hall sum of 12 uniforms, exact integer sum then one scale."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.int64)
    for lane in range(12):
        acc += u24(seed, name, n, lane=lane + 1)
    x = acc.astype(np.float64) / float(1 << 24) - 6.0
    return x.astype(np.float32).reshape(shape)


def randint(seed: int, name: str, shape: Sequence[int], lo: int, hi: int) -> np.ndarray:
    """Integers in [lo, hi)."""
    n = int(np.prod(shape))
    r = raw_bits(seed, name, n) >> np.uint64(11)
    return (lo + (r % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)


# --------------------------------------------------------------------------
# model configuration + state_dict
# --------------------------------------------------------------------------
def model_config(vocab_size: int = 512, embedding_dim: int = 256, hidden_dim: int = 256,
                 lstm_layers: int = 1, attention: bool = False, channels: int = 3,
                 img_height: int = 64, img_width: int = 320,
                 conv_filters: Sequence[int] = (32, 64, 128), kernel_size: int = 3,
                 pool_size: int = 2, dropout: float = 0.0, max_seq_length: int = 150) -> Dict:
    """The 'primary' dims of SURVEY.md section 8 by default."""
    return dict(model_type="cnn_lstm", vocab_size=vocab_size, embedding_dim=embedding_dim,
                hidden_dim=hidden_dim, lstm_layers=lstm_layers, attention=attention,
                channels=channels, img_height=img_height, img_width=img_width,
                conv_filters=list(conv_filters), kernel_size=kernel_size,
                pool_size=pool_size, dropout=dropout, max_seq_length=max_seq_length)


def encoder_params(cfg: Dict) -> Dict:
    return dict(img_height=cfg["img_height"], img_width=cfg["img_width"],
                channels=cfg["channels"], conv_filters=list(cfg["conv_filters"]),
                kernel_size=cfg["kernel_size"], pool_size=cfg["pool_size"],
                padding="same", embedding_dim=cfg["embedding_dim"])


def decoder_params(cfg: Dict) -> Dict:
    return dict(hidden_dim=cfg["hidden_dim"], max_seq_length=cfg["max_seq_length"],
                lstm_layers=cfg["lstm_layers"], dropout=cfg["dropout"],
                attention=cfg["attention"])


def flat_features(cfg: Dict) -> int:
    h, w = cfg["img_height"], cfg["img_width"]
    for _ in cfg["conv_filters"]:
        h, w = h // cfg["pool_size"], w // cfg["pool_size"]
    return cfg["conv_filters"][-1] * h * w


def make_state_dict(cfg: Dict, seed: int = 42, out_scale: float = 1.0,
                    end_bias: Optional[float] = None, enc_scale: float = 1.0,
                    end_row_scale: float = 1.0,
                    end_clock: Optional[Sequence[float]] = None) -> Dict[str, np.ndarray]:
    """float32 numpy state_dict with the reference's key names.

    ``out_scale`` multiplies decoder.output_layer.weight (wide argmax margins,
    SURVEY.md section 7 'hard parts'); ``end_bias`` if given is added to
    output_layer.bias[END] so that sequences terminate; ``enc_scale`` multiplies
    encoder.embedding_layer.weight so the image embedding is O(1) and rows decode
    to different sequences; ``end_row_scale`` multiplies output_layer.weight[END]
    so that END wins sporadically (state-dependent) rather than always/never;
    ``end_clock=(delta, gain, offset)`` turns hidden unit 0 of the top LSTM layer into
    a step counter (i,f,o gates saturated open, g = tanh(delta), no input weights, so
    c0 = t*delta) and wires it to the END logit (+gain*h0 - offset): END is unlikely
    early and certain late, at a row-dependent step.
    """
    sd: Dict[str, np.ndarray] = {}
    k = cfg["kernel_size"]
    cin = cfg["channels"]
    for i, cout in enumerate(cfg["conv_filters"]):
        bound = 1.0 / math.sqrt(cin * k * k)
        key = f"encoder.cnn_layers.{3 * i}"
        sd[key + ".weight"] = uniform(seed, key + ".weight", (cout, cin, k, k), -bound, bound)
        sd[key + ".bias"] = uniform(seed, key + ".bias", (cout,), -bound, bound)
        cin = cout
    E, H, V, L = cfg["embedding_dim"], cfg["hidden_dim"], cfg["vocab_size"], cfg["lstm_layers"]
    flat = flat_features(cfg)
    bound = 1.0 / math.sqrt(flat)
    sd["encoder.embedding_layer.weight"] = (uniform(seed, "encoder.embedding_layer.weight", (E, flat), -bound, bound)
                                            * np.float32(enc_scale)).astype(np.float32)
    sd["encoder.embedding_layer.bias"] = uniform(seed, "encoder.embedding_layer.bias", (E,), -bound, bound)
    sd["decoder.embedding.weight"] = normal_like(seed, "decoder.embedding.weight", (V, E))
    bound = 1.0 / math.sqrt(H)
    for l in range(L):
        in_l = 2 * E if l == 0 else H
        for nm, shp in ((f"weight_ih_l{l}", (4 * H, in_l)), (f"weight_hh_l{l}", (4 * H, H)),
                        (f"bias_ih_l{l}", (4 * H,)), (f"bias_hh_l{l}", (4 * H,))):
            sd["decoder.lstm." + nm] = uniform(seed, "decoder.lstm." + nm, shp, -bound, bound)
    if cfg["attention"]:
        bound = 1.0 / math.sqrt(H + E)
        sd["decoder.attention.attn.weight"] = uniform(seed, "decoder.attention.attn.weight", (H, H + E), -bound, bound)
        sd["decoder.attention.attn.bias"] = uniform(seed, "decoder.attention.attn.bias", (H,), -bound, bound)
        bound = 1.0 / math.sqrt(H)
        sd["decoder.attention.v.weight"] = uniform(seed, "decoder.attention.v.weight", (1, H), -bound, bound)
    bound = 1.0 / math.sqrt(H)
    w = uniform(seed, "decoder.output_layer.weight", (V, H), -bound, bound)
    w = (w * np.float32(out_scale)).astype(np.float32)
    w[END] = (w[END] * np.float32(end_row_scale)).astype(np.float32)
    sd["decoder.output_layer.weight"] = w
    b = uniform(seed, "decoder.output_layer.bias", (V,), -bound, bound)
    if end_bias is not None:
        b[END] += np.float32(end_bias)
    if end_clock is not None:
        delta, gain, offset = end_clock
        l = L - 1
        for g_idx, bias in ((0, 10.0), (1, 10.0), (2, float(delta)), (3, 10.0)):
            r = g_idx * H                                  # unit 0 of gate g_idx (order i,f,g,o)
            sd[f"decoder.lstm.weight_ih_l{l}"][r, :] = 0.0
            sd[f"decoder.lstm.weight_hh_l{l}"][r, :] = 0.0
            sd[f"decoder.lstm.bias_ih_l{l}"][r] = np.float32(bias)
            sd[f"decoder.lstm.bias_hh_l{l}"][r] = 0.0
        w[END, 0] = np.float32(gain)
        b[END] -= np.float32(offset)
    sd["decoder.output_layer.bias"] = b
    return sd


def make_images(batch: int, cfg: Dict, seed: int = 1234) -> np.ndarray:
    """float32 (B,C,H,W) NCHW, U[-1,1) (load_image's gray range, data/utils.py:73-75)."""
    return uniform(seed, "images", (batch, cfg["channels"], cfg["img_height"], cfg["img_width"]), -1.0, 1.0)


def make_formulas(batch: int, length: int, vocab_size: int, seed: int = 777,
                  min_len: int = 20) -> np.ndarray:
    """int64 (B,length): START, body U{4..V-1}, END at l~U{min_len..length-1}, PAD after."""
    body = randint(seed, "formula.body", (batch, length), 4, vocab_size)
    lo = min(min_len, length - 1)
    ends = randint(seed, "formula.len", (batch,), lo, length)
    out = body.copy()
    out[:, 0] = START
    col = np.arange(length)[None, :]
    out[col == ends[:, None]] = END
    out[col > ends[:, None]] = PAD
    return out.astype(np.int64)


def checksum(a: np.ndarray) -> float:
    """Order-sensitive float64 checksum used by fixtures."""
    f = np.asarray(a, dtype=np.float64).ravel()
    w = (np.arange(f.size, dtype=np.float64) % 251.0) + 1.0
    return float(np.dot(f, w))


def make_resnet_state_dict(names_and_shapes, seed: int = 42) -> Dict[str, np.ndarray]:
    """Deterministic ResNet weights for a given list of (state_dict key, shape): conv ~N(0, 2/fan_out)
    (torchvision's init), BatchNorm with NON-trivial statistics (so that BN folding is exercised),
    Linear U(+-1/sqrt(fan_in))."""
    sd: Dict[str, np.ndarray] = {}
    for name, shape in names_and_shapes:
        shape = tuple(shape)
        if name.endswith("num_batches_tracked"):
            sd[name] = np.zeros(shape, dtype=np.int64)
        elif name.endswith("running_var"):
            sd[name] = uniform(seed, name, shape, 0.5, 1.5)
        elif name.endswith("running_mean"):
            sd[name] = uniform(seed, name, shape, -0.2, 0.2)
        elif len(shape) == 4:
            fan_out = shape[0] * shape[2] * shape[3]
            sd[name] = (normal_like(seed, name, shape) * np.float32(math.sqrt(2.0 / fan_out))).astype(np.float32)
        elif len(shape) == 2:
            b = 1.0 / math.sqrt(shape[1])
            sd[name] = uniform(seed, name, shape, -b, b)
        elif name.endswith(".weight"):                       # BatchNorm gamma
            sd[name] = uniform(seed, name, shape, 0.5, 1.5)
        else:                                                # biases (BatchNorm beta, Linear bias)
            sd[name] = uniform(seed, name, shape, -0.2, 0.2)
    return sd
