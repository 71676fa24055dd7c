"""CPU oracle for the img2latex hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A functional, fp32, torch-CPU restatement of the reference algorithm
(Jeremy-Cleland/hmer-img2latex @ 2025-04-18), operating on a plain
``state_dict`` (name -> tensor) with the reference's key names.  Each function
cites the reference file:line it follows.  Only ``tests/``, ``bench.py``'s
``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import this module; the
product package (``hmer-img2latex_amd/``) never does, and has no CPU fallback.

Parity status: PINNED for the CNN-LSTM path -- ``tests/golden/*.npz`` were
produced by importing the real reference in the build container
(``tests/golden/make_golden.py``) and ``tests/test_oracle_golden.py`` checks this
restatement against them (token ids exact, floats <= 1e-6).
ResNet encoder: parity UNPINNED (torchvision absent, remote weights).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
Hidden = Tuple[torch.Tensor, torch.Tensor]


def to_torch_sd(np_sd) -> SD:
    return {k: torch.from_numpy(v.copy()) for k, v in np_sd.items()}


# --------------------------------------------------------------------------
# encoder  (reference img2latex/model/encoder.py)
# --------------------------------------------------------------------------
def conv_block(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, pool: int = 2) -> torch.Tensor:
    """Conv2d(k, padding=k//2) -> ReLU -> MaxPool2d(pool)   (encoder.py:72,78-93)."""
    k = w.shape[-1]
    return F.max_pool2d(F.relu(F.conv2d(x, w, b, padding=k // 2)), pool)


def pool_windows(conv: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) -> (B,C,H/2,W/2,4): the four values of every 2x2 pooling window, index 2*dy + dx (floor pooling)."""
    B, C, H, W = conv.shape
    c = conv[:, :, : H // 2 * 2, : W // 2 * 2]
    return c.reshape(B, C, H // 2, 2, W // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, H // 2, W // 2, 4)


def conv_block_decided(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, argmax: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """conv_block with the two DISCRETE choices of the block imposed instead of evaluated: `argmax` (which of the four
    window values is pooled) and `gate` (whether the pooled value passes the ReLU).  The result is the same smooth
    function of (x, w, b) that the block is wherever those choices are the block's own -- test tooling to separate an
    implementation's rounding of the smooth part from its resolution of near-ties (tests/test_hip_training.py)."""
    win = pool_windows(F.conv2d(x, w, b, padding=w.shape[-1] // 2))
    return win.gather(-1, argmax.long().unsqueeze(-1)).squeeze(-1) * gate.to(win.dtype)


def cnn_blocks(sd: SD, cfg: Dict, x: torch.Tensor, decisions=None) -> List[torch.Tensor]:
    """Outputs of every conv block (Sequential idx 0,3,6 hold the convs, encoder.py:78-95).
    decisions: None, or per block (argmax, gate) for conv_block_decided."""
    outs = []
    for i in range(len(cfg["conv_filters"])):
        w, b = sd[f"encoder.cnn_layers.{3 * i}.weight"], sd[f"encoder.cnn_layers.{3 * i}.bias"]
        x = conv_block(x, w, b, cfg["pool_size"]) if decisions is None else conv_block_decided(x, w, b, *decisions[i])
        outs.append(x)
    return outs


def cnn_encoder(sd: SD, cfg: Dict, x: torch.Tensor, decisions=None) -> torch.Tensor:
    """CNNEncoder.forward: blocks -> Flatten (NCHW row-major) -> Linear -> ReLU (encoder.py:111-129)."""
    feat = cnn_blocks(sd, cfg, x, decisions)[-1].flatten(1)
    return F.relu(F.linear(feat, sd["encoder.embedding_layer.weight"],
                           sd["encoder.embedding_layer.bias"]))


# --------------------------------------------------------------------------
# decoder  (reference img2latex/model/decoder.py)
# --------------------------------------------------------------------------
def attention_context(sd: SD, h_top: torch.Tensor, enc: torch.Tensor) -> torch.Tensor:
    """Attention.forward with encoder_outputs (B,1,E)  (decoder.py:312-343).

    h_top (B,1,H), enc (B,1,E) -> context (B,1,E).  src_len == 1, so the softmax
    is over ONE element and the context equals enc bit-for-bit.
    """
    src_len = enc.shape[1]
    hid = h_top.repeat(1, src_len, 1)                                   # :329
    energy = torch.tanh(F.linear(torch.cat((hid, enc), dim=2),          # :332
                                 sd["decoder.attention.attn.weight"],
                                 sd["decoder.attention.attn.bias"]))
    score = F.linear(energy, sd["decoder.attention.v.weight"]).squeeze(2)  # :335
    wts = F.softmax(score, dim=1).unsqueeze(1)                          # :338
    return torch.bmm(wts, enc)                                          # :341


def lstm_step(sd: SD, cfg: Dict, x: torch.Tensor, h: torch.Tensor, c: torch.Tensor,
              dropout_p: float = 0.0, training: bool = False) -> Tuple[torch.Tensor, Hidden]:
    """One time step of nn.LSTM(2E->H, L layers) (decoder.py:76-82), gate order i,f,g,o.

    x (B, 2E); h, c (L,B,H).  Returns (top-layer h (B,H), (h', c')).
    Inter-layer dropout applies only in training with L > 1 (decoder.py:81).
    """
    H = cfg["hidden_dim"]
    hs, cs = [], []
    inp = x
    for l in range(cfg["lstm_layers"]):
        gates = (F.linear(inp, sd[f"decoder.lstm.weight_ih_l{l}"], sd[f"decoder.lstm.bias_ih_l{l}"])
                 + F.linear(h[l], sd[f"decoder.lstm.weight_hh_l{l}"], sd[f"decoder.lstm.bias_hh_l{l}"]))
        i, f, g, o = gates.split(H, dim=1)
        c_new = torch.sigmoid(f) * c[l] + torch.sigmoid(i) * torch.tanh(g)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        hs.append(h_new)
        cs.append(c_new)
        inp = h_new
        if training and dropout_p > 0 and l < cfg["lstm_layers"] - 1:
            inp = F.dropout(inp, dropout_p, True)
    return inp, (torch.stack(hs), torch.stack(cs))


def zero_hidden(cfg: Dict, batch: int) -> Hidden:
    z = torch.zeros(cfg["lstm_layers"], batch, cfg["hidden_dim"])
    return z, z.clone()


def decode_step(sd: SD, cfg: Dict, enc: torch.Tensor, tok: torch.Tensor,
                hidden: Optional[Hidden]) -> Tuple[torch.Tensor, Hidden]:
    """LSTMDecoder.decode_step (decoder.py:197-284).

    enc (B,E), tok (B,1) int64, hidden (h,c) each (L,B,H) or None ->
    (logits (B,1,V), (h',c')).  No dropout on this path.
    """
    B = tok.shape[0]
    emb = F.embedding(tok, sd["decoder.embedding.weight"])              # :214 (B,1,E)
    if hidden is None:                                                  # :231-244 / :253-266
        hidden = zero_hidden(cfg, B)
    h, c = hidden
    if cfg["attention"]:
        ctx = attention_context(sd, h[-1].unsqueeze(1), enc.unsqueeze(1))   # :271
    else:
        ctx = enc.unsqueeze(1)                                          # :218
    x = torch.cat([emb, ctx], dim=2).squeeze(1)                         # :228 / :274
    top, hidden = lstm_step(sd, cfg, x, h, c)                           # :247 / :277
    logits = F.linear(top, sd["decoder.output_layer.weight"], sd["decoder.output_layer.bias"])
    return logits.unsqueeze(1), hidden                                  # :250 / :280


def decoder_forward(sd: SD, cfg: Dict, enc: torch.Tensor, target: torch.Tensor,
                    dropout_p: float = 0.0, training: bool = False) -> torch.Tensor:
    """LSTMDecoder.forward, teacher forcing (decoder.py:100-195) -> (B,T,V).

    no-attn: dropout(cat[emb, enc]) -> LSTM -> dropout -> Linear (:121-143)
    attn:    dropout(emb) then per step [attention, cat, LSTM, dropout, Linear] (:144-193)
    With dropout_p == 0 / eval both paths compute the same numbers.
    """
    B, T = target.shape
    emb = F.embedding(target, sd["decoder.embedding.weight"])
    h, c = zero_hidden(cfg, B)
    outs = []
    if cfg["attention"]:
        emb = F.dropout(emb, dropout_p, training)                       # :162
    for t in range(T):
        if cfg["attention"]:
            ctx = attention_context(sd, h[-1].unsqueeze(1), enc.unsqueeze(1)).squeeze(1)
            x = torch.cat([emb[:, t, :], ctx], dim=1)
        else:
            x = F.dropout(torch.cat([emb[:, t, :], enc], dim=1), dropout_p, training)  # :130-133
        top, (h, c) = lstm_step(sd, cfg, x, h, c, dropout_p, training)
        top = F.dropout(top, dropout_p, training)                       # :139 / :186
        outs.append(F.linear(top, sd["decoder.output_layer.weight"],
                             sd["decoder.output_layer.bias"]))
    return torch.stack(outs, dim=1)


# --------------------------------------------------------------------------
# seq2seq  (reference img2latex/model/seq2seq.py)
# --------------------------------------------------------------------------
def seq2seq_forward(sd: SD, cfg: Dict, images: torch.Tensor, formulas: torch.Tensor,
                    dropout_p: float = 0.0, training: bool = False, decisions=None) -> torch.Tensor:
    """Seq2SeqModel.forward: decoder(encoder(images), formulas[:, :-1]) (seq2seq.py:98-122)."""
    return decoder_forward(sd, cfg, cnn_encoder(sd, cfg, images, decisions), formulas[:, :-1],
                           dropout_p, training)


def greedy_search(sd: SD, cfg: Dict, enc: torch.Tensor, start_id: int, end_id: int,
                  max_length: int, temperature: float = 1.0, return_margins: bool = False):
    """Seq2SeqModel._greedy_search (seq2seq.py:192-232).

    argmax of logits (/temperature if != 1); stops only when ALL rows emit END in
    the same step (:220).  B == 1: strip START, truncate at END (:224-231);
    B > 1: raw lists including START.
    """
    B = enc.shape[0]
    tok = torch.full((B, 1), start_id, dtype=torch.long)
    hidden = None
    seqs = [[start_id] for _ in range(B)]
    margins = []
    for _ in range(max_length):
        out, hidden = decode_step(sd, cfg, enc, tok, hidden)
        logits = out.squeeze(1)
        if temperature != 1.0:
            logits = logits / temperature
        nxt = torch.argmax(logits, dim=-1)
        if return_margins:
            top2 = torch.topk(logits, 2, dim=-1).values
            margins.append((top2[:, 0] - top2[:, 1]).clone())
        tok = nxt.unsqueeze(1)
        vals = nxt.tolist()
        for i in range(B):
            seqs[i].append(vals[i])
        if all(v == end_id for v in vals):
            break
    res = seqs
    if B == 1:
        s = seqs[0]
        if s and s[0] == start_id:
            s = s[1:]
        if end_id in s:
            s = s[: s.index(end_id)]
        res = s
    if return_margins:
        return res, torch.stack(margins, dim=1)
    return res


def beam_search(sd: SD, cfg: Dict, enc: torch.Tensor, start_id: int, end_id: int,
                max_length: int, beam_size: int, return_score: bool = False):
    """Seq2SeqModel._beam_search (seq2seq.py:234-298); B must be 1 else greedy (:244-247).

    Per live beam: decode_step at B=1 -> log_softmax (fp32) -> topk(k) -> candidates
    with score + log_p accumulated in Python float (fp64) (:266-275); ended beams
    move to ``completed`` on the NEXT iteration (:258-260); stable sort desc, keep k
    (:279-280); early exit when all k ended (:282-284); best = max(completed) (first
    on ties) else beams[0] (:286-290); strip START / cut at END (:291-297).
    """
    if enc.shape[0] != 1:
        return greedy_search(sd, cfg, enc, start_id, end_id, max_length, 1.0)
    beams = [dict(tokens=[start_id], hidden=None, score=0.0)]
    completed = []
    for _ in range(max_length):
        cands = []
        for bm in beams:
            last = bm["tokens"][-1]
            if last == end_id:
                completed.append(bm)
                continue
            out, hid = decode_step(sd, cfg, enc, torch.tensor([[last]], dtype=torch.long), bm["hidden"])
            logp = torch.log_softmax(out.squeeze(1), dim=-1).squeeze(0)
            tv, ti = torch.topk(logp, beam_size)
            for lp, ix in zip(tv.tolist(), ti.tolist()):
                cands.append(dict(tokens=bm["tokens"] + [ix], hidden=hid, score=bm["score"] + lp))
        if not cands:
            break
        cands = sorted(cands, key=lambda b: b["score"], reverse=True)
        beams = cands[:beam_size]
        if all(b["tokens"][-1] == end_id for b in beams):
            completed.extend(beams)
            break
    best = max(completed, key=lambda b: b["score"]) if completed else beams[0]
    seq = best["tokens"]
    if seq and seq[0] == start_id:
        seq = seq[1:]
    if end_id in seq:
        seq = seq[: seq.index(end_id)]
    if return_score:
        return seq, best["score"]
    return seq


def inference(sd: SD, cfg: Dict, image: torch.Tensor, start_id: int, end_id: int,
              max_length: Optional[int] = None, temperature: Optional[float] = None,
              beam_size: Optional[int] = None):
    """Seq2SeqModel.inference dispatch and defaults (seq2seq.py:124-190)."""
    max_length = 150 if max_length is None else max_length
    temperature = 1.0 if temperature is None else temperature
    beam_size = 0 if beam_size is None else beam_size
    enc = cnn_encoder(sd, cfg, image)
    if beam_size > 0:
        return beam_search(sd, cfg, enc, start_id, end_id, max_length, beam_size)
    return greedy_search(sd, cfg, enc, start_id, end_id, max_length, temperature)


# --------------------------------------------------------------------------
# Predictor.predict_batch greedy loop  (reference img2latex/training/predictor.py)
# --------------------------------------------------------------------------
def predictor_greedy_loop(sd: SD, cfg: Dict, enc: torch.Tensor, start_id: int, end_id: int,
                          max_length: int, temperature: float = 1.0, top_k: int = 0,
                          top_p: float = 0.0, generator: Optional[torch.Generator] = None
                          ) -> List[List[int]]:
    """predictor.py:264-358: argmax of softmax(logits/T), sticky ``finished`` flags,
    stop when all finished, trim each row at its first END (START kept).
    top-k / top-p masks restated from :299-327; sampling (:330-331) uses
    torch.multinomial and cannot be bit-matched across implementations.
    """
    B = enc.shape[0]
    seqs = torch.full((B, 1), start_id, dtype=torch.long)
    finished = torch.zeros(B, dtype=torch.bool)
    hidden = None
    for _ in range(max_length):
        out, hidden = decode_step(sd, cfg, enc, seqs[:, -1].unsqueeze(1), hidden)
        logits = out.squeeze(1)
        if temperature != 1.0:
            logits = logits / temperature
        probs = torch.softmax(logits, dim=-1)
        if top_k > 0:
            top_k = min(top_k, probs.size(-1))
            kth = torch.topk(probs, top_k, dim=-1).values[:, -1, None]
            probs = torch.where(probs < kth, torch.zeros_like(probs), probs)
            s = probs.sum(dim=-1, keepdim=True)
            if torch.any(s > 0):
                probs = probs / s
        if top_p > 0.0:
            sp, si = torch.sort(probs, descending=True)
            cum = torch.cumsum(sp, dim=-1)
            rm = cum > top_p
            rm[:, 1:] = rm[:, :-1].clone()
            rm[:, 0] = False
            mask = rm.scatter(-1, si, rm)
            probs = torch.where(mask, torch.zeros_like(probs), probs)
            s = probs.sum(dim=-1, keepdim=True)
            if torch.any(s > 0):
                probs = probs / s
        if temperature > 0 and (top_k > 0 or top_p > 0.0):
            nxt = torch.multinomial(probs, 1, generator=generator)
        else:
            nxt = torch.argmax(probs, dim=-1, keepdim=True)
        seqs = torch.cat([seqs, nxt], dim=1)
        finished = finished | (nxt.squeeze(1) == end_id)
        if bool(torch.all(finished)):
            break
    res = []
    for row in seqs.tolist():
        res.append(row[: row.index(end_id)] if end_id in row else row)
    return res


# --------------------------------------------------------------------------
# training step  (reference img2latex/training/trainer.py)
# --------------------------------------------------------------------------
def ce_label_smooth(logits: torch.Tensor, targets: torch.Tensor, pad_id: int = 0,
                    eps: float = 0.1) -> torch.Tensor:
    """CrossEntropyLoss(ignore_index=PAD, reduction='mean', label_smoothing=0.1) on
    logits.transpose(1,2) (trainer.py:111-115,335-336), written out:
    mean over non-PAD tokens of (1-eps)*nll + eps*(-mean_v logp).
    """
    logp = torch.log_softmax(logits, dim=-1)                            # (B,T,V)
    nll = -logp.gather(-1, targets.unsqueeze(-1)).squeeze(-1)
    smooth = -logp.mean(dim=-1)
    keep = targets != pad_id
    per_tok = (1.0 - eps) * nll + eps * smooth
    return (per_tok * keep).sum() / keep.sum()


def clip_grad_norm(grads: Dict[str, torch.Tensor], max_norm: float) -> torch.Tensor:
    """nn.utils.clip_grad_norm_ (trainer.py:338-341): scale by max_norm/(total+1e-6), clamped to 1."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def adam_step(sd: SD, grads: Dict[str, torch.Tensor], state: Dict, lr: float = 1e-3,
              weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """optim.Adam(lr, weight_decay) single-tensor update (trainer.py:91-93): L2 is
    COUPLED (added to the gradient), bias-corrected, denom = sqrt(v)/sqrt(bc2) + eps."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    for k, p in sd.items():
        g = grads[k]
        if weight_decay != 0:
            g = g + weight_decay * p
        m = state.setdefault("m." + k, torch.zeros_like(p))
        v = state.setdefault("v." + k, torch.zeros_like(p))
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def loss_and_grads(sd: SD, cfg: Dict, images: torch.Tensor, formulas: torch.Tensor, pad_id: int = 0,
                   dtype: torch.dtype = torch.float32, decisions=None):
    """Loss and d(loss)/d(parameter) of the training forward (trainer.py:334-337), dropout off, evaluated in ``dtype``.
    float64 gives the reference point for judging fp32 results: the pooling arg max and the ReLU boundary make the
    conv gradients discontinuous, so two correct fp32 evaluations differ by ~1e-3 of a gradient's maximum.
    ``decisions`` (per conv block (argmax, gate), see conv_block_decided) imposes an implementation's discrete choices,
    which leaves a smooth function whose float64 gradient that implementation must match closely."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        params = {k: v.detach().to(dtype).requires_grad_(True) for k, v in sd.items()}
        logits = seq2seq_forward(params, cfg, images.to(dtype), formulas, decisions=decisions)
        loss = ce_label_smooth(logits, formulas[:, 1:], pad_id)
        gl = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    finally:
        torch.set_default_dtype(old)
    grads = {k: (torch.zeros_like(params[k]) if g is None else g.detach()) for k, g in zip(params.keys(), gl)}
    return float(loss.detach()), grads


def train_step(sd: SD, cfg: Dict, images: torch.Tensor, formulas: torch.Tensor, state: Dict,
               lr: float = 1e-3, weight_decay: float = 1e-4, clip: float = 5.0,
               pad_id: int = 0) -> Dict:
    """Trainer.train_epoch fp32 branch, one batch (trainer.py:303-343), dropout off.

    targets = formulas[:, 1:] (:306); loss (:334-336); backward (:337); clip (:338-341);
    Adam step (:342).  Updates ``sd`` in place; returns loss / grads / total norm.
    """
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    logits = seq2seq_forward(params, cfg, images, formulas)
    loss = ce_label_smooth(logits, formulas[:, 1:], pad_id)
    gl = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    grads = {k: (torch.zeros_like(sd[k]) if g is None else g.detach().clone())
             for k, g in zip(params.keys(), gl)}
    raw = {k: g.clone() for k, g in grads.items()}
    total = clip_grad_norm(grads, clip) if clip > 0 else torch.tensor(0.0)
    with torch.no_grad():
        adam_step(sd, grads, state, lr, weight_decay)
    return dict(loss=float(loss.detach()), total_norm=float(total), grads=raw, logits=logits.detach())
