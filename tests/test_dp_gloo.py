"""Data-parallel host logic on CPU: world_size 2, gloo.  Each rank computes the gradient of the SUM
loss of its shard (with the CPU oracle, the only CPU arithmetic available to tests), the product's
dp helpers all-reduce one flat buffer [grads..., loss_sum, count] and divide by the GLOBAL count;
the result must equal the single-process full-batch gradient of the reference's mean loss."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import img2latex_oracle as O
from helpers import PAD
from img2latex_amd import synth
from img2latex_amd.training.dp import all_reduce_gradients, broadcast_parameters, shard_batch

CFG = dict(vocab_size=50, embedding_dim=32, hidden_dim=64, lstm_layers=1, attention=False, channels=1,
           img_height=16, img_width=32, conv_filters=(4, 8, 16))
B, T = 7, 10                      # 7 rows over 2 ranks: ragged shards (4 + 3)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _sum_loss_grads(sd, cfg, x, forms):
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits = O.seq2seq_forward(params, cfg, x, forms)
    logp = torch.log_softmax(logits, dim=-1)
    tgt = forms[:, 1:]
    keep = tgt != PAD
    per = 0.9 * -logp.gather(-1, tgt.unsqueeze(-1)).squeeze(-1) + 0.1 * -logp.mean(-1)
    loss_sum = (per * keep).sum()
    loss_sum.backward()
    flat = torch.cat([params[k].grad.reshape(-1) for k in sd] + [loss_sum.detach().reshape(1), keep.sum().float().reshape(1)])
    return flat


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = synth.model_config(**CFG)
    sd = O.to_torch_sd(synth.make_state_dict(cfg, seed=3 + rank))          # ranks start DIFFERENT on purpose
    flat_p = torch.cat([v.reshape(-1) for v in sd.values()])
    broadcast_parameters(flat_p, src=0)                                     # ... and are made identical
    off = 0
    for k in sd:
        n = sd[k].numel()
        sd[k] = flat_p[off:off + n].view_as(sd[k]).clone()
        off += n
    x = torch.from_numpy(synth.make_images(B, cfg, seed=1234))
    forms = torch.from_numpy(synth.make_formulas(B, T, cfg["vocab_size"], seed=777, min_len=4))
    lo, hi = shard_batch(B, rank, world)
    flat = _sum_loss_grads(sd, cfg, x[lo:hi], forms[lo:hi])
    all_reduce_gradients(flat)
    grads = flat[:-2] / flat[-1]
    if rank == 0:
        full = _sum_loss_grads(sd, cfg, x, forms)
        want = full[:-2] / full[-1]
        out["max_err"] = float((grads - want).abs().max())
        out["scale"] = float(want.abs().max())
        out["count"] = (float(flat[-1]), float(full[-1]))
        out["loss"] = (float(flat[-2] / flat[-1]), float(full[-2] / full[-1]))
    gathered = [torch.zeros_like(grads) for _ in range(world)]
    dist.all_gather(gathered, grads)
    if rank == 0:
        out["identical_across_ranks"] = bool(torch.equal(gathered[0], gathered[1]))
    dist.destroy_process_group()


def test_shard_batch_covers_everything():
    for n in (1, 7, 64, 512):
        for world in (1, 2, 3, 8):
            spans = [shard_batch(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_allreduce_equals_full_batch():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out["count"][0] == out["count"][1]
    assert abs(out["loss"][0] - out["loss"][1]) < 1e-6
    assert out["max_err"] <= 1e-6 * max(1.0, out["scale"]), dict(out)
    assert out["identical_across_ranks"]


def _overlap_worker(rank, world, port, out):
    from img2latex_amd.training.dp import OverlappedAllReduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, split = 10007, 9000
    base = torch.from_numpy(synth.uniform(50 + rank, "flat", (n,), -1.0, 1.0))
    a, b, c = base.clone(), base.clone(), base.clone()
    all_reduce_gradients(a)                                  # the single all-reduce
    r = OverlappedAllReduce(b, split)
    r.start_early()                                          # early piece in flight ...
    b[split:] += 0.0                                         # ... while "conv backward" still writes the late piece
    r.finish()
    r2 = OverlappedAllReduce(c, split)                       # finish() alone = one all-reduce of everything
    r2.finish()
    if rank == 0:
        out["two_piece_equals_single"] = bool(torch.equal(a, b))
        out["finish_only_equals_single"] = bool(torch.equal(a, c))
    dist.destroy_process_group()


def test_two_piece_allreduce_is_bit_identical_to_one():
    """dp.OverlappedAllReduce: [decoder + FC gradients] started early, [conv gradients, loss sum, count] at the end:
    element-wise the same sums as ONE all-reduce of the flat buffer (world size 2, gloo)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out["two_piece_equals_single"] and out["finish_only_equals_single"]


def _step_worker(rank, world, port, out, overlap):
    """One full optimisation step the way TrainStep does it under data parallelism, on the CPU oracle's arithmetic: shard ->
    SUM-loss gradients into one flat buffer [grads..., loss_sum, count] -> the product's all-reduce (one call, or
    dp.OverlappedAllReduce's two pieces) -> division by the GLOBAL count -> clip by the GLOBAL norm -> Adam."""
    from img2latex_amd.training.dp import OverlappedAllReduce
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg = synth.model_config(**CFG)
    sd = O.to_torch_sd(synth.make_state_dict(cfg, seed=3))
    n_rows = 509                                                    # 8 x 63 + 5: ragged shards (64 x 5, 63 x 3)
    x = torch.from_numpy(synth.make_images(n_rows, cfg, seed=1234))
    forms = torch.from_numpy(synth.make_formulas(n_rows, T, cfg["vocab_size"], seed=777, min_len=4))
    lo, hi = shard_batch(n_rows, rank, world)
    flat = _sum_loss_grads(sd, cfg, x[lo:hi], forms[lo:hi])
    if overlap:                     # two pieces, as TrainStep issues them (its split sits in front of the conv gradients;
        r = OverlappedAllReduce(flat, (2 * flat.numel()) // 3)      # element-wise sums do not care where)
        r.start_early()
        r.finish()
    else:
        all_reduce_gradients(flat)
    count = flat[-1]
    grads, off = {}, 0
    for k, v in sd.items():
        grads[k] = (flat[off:off + v.numel()] / count).view_as(v).clone()
        off += v.numel()
    total = O.clip_grad_norm(grads, 5.0)
    state = {}
    with torch.no_grad():
        O.adam_step(sd, grads, state)
    mine = torch.cat([v.reshape(-1) for v in sd.values()])
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank == 0:
        ref = O.to_torch_sd(synth.make_state_dict(cfg, seed=3))
        res = O.train_step(ref, cfg, x, forms, {})
        want = torch.cat([v.reshape(-1) for v in ref.values()])
        out["param_err"] = float((mine - want).abs().max())
        out["norm"] = (float(total), res["total_norm"])
        out["loss"] = (float(flat[-2] / count), res["loss"])
        out["count"] = float(count)
        out["replicas_identical"] = all(bool(torch.equal(g, gathered[0])) for g in gathered)
        out["shards"] = [shard_batch(n_rows, r_, world) for r_ in range(world)]
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_eight_rank_step_equals_single_process_oracle_step(overlap):
    """The N = 8 case the driver runs on a whole node, rehearsed on the CPU (gloo, world size 8): 509 ragged rows -> 8
    shards, one flat-buffer all-reduce (or its two pieces), global-count division, clip by the global norm, Adam: every
    replica ends with the SAME parameters, within 1e-5 of the single-process oracle step on the full batch.
    When an 8-GPU node exists the only new thing it exercises is RCCL itself."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_step_worker, args=(8, _free_port(), out, overlap), nprocs=8, join=True)
    sizes = [hi - lo for lo, hi in out["shards"]]
    assert sum(sizes) == 509 and sorted(set(sizes)) == [63, 64]
    assert out["replicas_identical"]
    assert abs(out["loss"][0] - out["loss"][1]) <= 1e-5 * max(1.0, abs(out["loss"][1])), dict(out)
    assert abs(out["norm"][0] - out["norm"][1]) <= 1e-5 * max(1.0, out["norm"][1]), dict(out)
    assert out["param_err"] <= 1e-5, dict(out)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_card_rehearsal():
    """`python bench.py --gpus 2` with no launcher around it (the self-launch path) as a one-card rehearsal: two ranks
    over gloo share the GPU, rank 0 prints ONE line with n_gpus = 2 and twice the per-rank batch; the train mode's line
    carries the all-reduce timing fields.  (RCCL needs one GPU per rank: the driver's multi-GPU run is the real thing.)"""
    import json
    import subprocess
    import sys
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["I2L_DIST_BACKEND"] = "gloo"
    ar = ("all_reduce_ms", "all_reduce_busbw_gbs", "all_reduce_payload_mb")
    for mode, keys, extra in (("greedy", (), []), ("train", ar, []), ("train", ar, ["--dp-single-allreduce"])):
        r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                            "--mode", mode, "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak"
        assert d["config"]["global_batch"] == 2 * d["config"]["batch_per_gpu"]
        for k in keys:
            assert k in d["config"], k
