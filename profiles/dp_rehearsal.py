"""Data-parallel rehearsal on ONE GPU box: N ranks (processes) share cuda:0, the collective backend is gloo (RCCL
refuses two ranks on one device), so what is exercised is TrainStep's real multi-process path -- parameter broadcast,
the two-piece overlapped gradient all-reduce of the flat buffer [grads, loss sum, count], division by the GLOBAL
count, clip + Adam on every rank -- against the single-process full-batch step (within 1e-5).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        profiles/dp_rehearsal.py
"""
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(REPO, "hmer-img2latex_amd"))
from img2latex_amd import synth  # noqa: E402
from img2latex_amd.model import Seq2SeqModel  # noqa: E402
from img2latex_amd.training import TrainStep, shard_batch  # noqa: E402


def model_for(cfg, seed):
    m = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=seed).items()})
    return m.to("cuda")


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.model_config(dropout=0.0)
    B, T = 64, 150
    x = torch.from_numpy(synth.make_images(B, cfg, seed=1234)).to("cuda")
    forms = torch.from_numpy(synth.make_formulas(B, T, cfg["vocab_size"], seed=777)).to("cuda")
    ts = TrainStep(model_for(cfg, 42 + rank))          # ranks start DIFFERENT: the constructor's broadcast fixes that
    full = TrainStep(model_for(cfg, 42))               # the reference step (its constructor broadcasts too: all ranks)
    full._reducer._active = lambda: False              # ... one process, the whole batch, no collective
    lo, hi = shard_batch(B, rank, world)
    losses, ref_losses, errs, same = [], [], [], True
    for _ in range(3):
        out = ts.step(x[lo:hi].contiguous(), forms[lo:hi].contiguous())
        losses.append(float(out["loss"]))
        mine = ts.flat_params.clone()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        if rank == 0:
            ref_losses.append(float(full.step(x, forms)["loss"]))
            same = same and all(torch.equal(gathered[0], g) for g in gathered[1:])
            errs.append(float((gathered[0] - full.flat_params).abs().max()))
    if rank == 0:
        print(f"ranks={world} replicas bit-identical: {same}; max |param - single-process| after steps 1..3: "
              f"{['%.2e' % e for e in errs]}; losses dp {['%.6f' % v for v in losses]} single {['%.6f' % v for v in ref_losses]}",
              flush=True)
        # within 1e-5 after ONE step (Adam's first steps amplify the summation-order difference afterwards)
        assert same and errs[0] <= 1e-5 and all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(losses, ref_losses))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
