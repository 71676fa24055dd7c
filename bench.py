#!/usr/bin/env python3
"""Headline benchmark: decoded LaTeX tokens/s, CNN-LSTM greedy decode, batch 256 per GPU,
320x64x3 synthetic images, 150 decode steps (BASELINE.json configs[1]).

    python bench.py                       # N = 1, 100 steps after 20 warm-up steps
    python bench.py --gpus N              # starts its N ranks itself (one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch already resident in HBM:
3 fused conv blocks -> Flatten+Linear+ReLU -> decoder prepare -> persistent greedy decode
loop of 150 steps -> token ids copied to (pinned) host memory.  Inference shards with no
collective (images are independent), so N GPUs run N replicas on different batches
("scaling": "weak"); value = all ranks' tokens / max-over-ranks time.

Timed region (default since r03): K batches through GreedyPipeline -- two HIP streams, the decode of batch i on a grouped
kernel with one wave per SIMD (r04 default: decode_group16_kernel, 16 members x 16 rows, per-step products as split-bf16
MFMA bursts; --decode-members 8: r03's decode_group8_kernel on the vector ALUs) while the conv workgroups of batch i + 1's
encoder run on the SAME CUs (the decode is bound by L2 round trips, the encoder by the matrix pipe: they share the chip by
resource).  encoder(i + 1) is held back until decode(i) reports itself resident (a dependency, not a delay).  The
`roofline` object describes THIS region's dominant kernel from HIP events recorded on the decode stream around each of
its K launches; `roofline.serial` is the serial pass's view, `schedule` says which schedule the two streams were in.  Every batch still runs the full encoder, prepare, 150 decode steps and the id copy, and
all K batches complete inside the timed region.  A second, SERIAL pass of K batches (one stream, each kernel alone on
the chip, the 4-member grouped decode) follows: it provides the per-kernel times of the roofline object, `value_serial`,
and a floor -- `value` is never worse than it.  --serial times only that pass (the r02 default).
Order of a default run: [W warm-up + K timed batches straight after start-up -> `value_cold_start`], [--settle-ms of the same
step, untimed: the GPU's clocks need ~50 ms of load after idle, profiles/r03/ramp.txt -> `config.settle`], [W warm-up +
EXACTLY K timed batches between barrier + synchronize fences -> `value`], [the serial pass], then the parity checks of both
regions' ids against the reference fixture.

Prints ONE JSON line (rank 0) with the driver's keys plus "roofline" (dominant kernel,
timed live with HIP events on the launch stream) and "cpu_baseline" (the CPU oracle on
this host's cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "hmer-img2latex_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from img2latex_amd import _lib, synth  # noqa: E402
from img2latex_amd.model import Seq2SeqModel  # noqa: E402
from img2latex_amd.pipeline import GreedyPipeline  # noqa: E402

# MI355X peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_HBM_GBS = 8000.0
PEAK_FP32_TFLOPS = 157.3          # fp32 MFMA = fp32 vector peak
PEAK_BF16_TFLOPS = 2500.0         # dense bf16 MFMA
SUSTAINED_BF16_TFLOPS = 1820.0     # bare MFMA loops with random operands (profiles/micro/mfma_shapes.hip): the clock drops under toggling data


def stage_costs(cfg, B, steps):
    """Algorithmic flops / bytes per launch, keyed by stage mark."""
    out = {}
    cin, h, w = cfg["channels"], cfg["img_height"], cfg["img_width"]
    for i, cout in enumerate(cfg["conv_filters"]):
        flops = 2.0 * 9 * cin * cout * h * w * B
        byts = 4.0 * (B * cin * h * w + B * cout * (h // 2) * (w // 2) + cout * cin * 9 + cout)
        out[f"conv{i}"] = dict(flops=flops, bytes=byts)
        cin, h, w = cout, h // 2, w // 2
    K, E, H, V = cin * h * w, cfg["embedding_dim"], cfg["hidden_dim"], cfg["vocab_size"]
    out["fc"] = dict(flops=2.0 * B * K * E, bytes=4.0 * (B * K + E * K + B * E + E))
    L = cfg["lstm_layers"]
    per_tok = 2.0 * (4 * H * (2 * E + H) + (L - 1) * 4 * H * 2 * H + H * V)      # 1.835 MFLOP at primary dims
    wbytes = 4.0 * (4 * H * H * (2 * L - 1) + H * V + V * 4 * H)                 # images streamed per step
    out["decode"] = dict(flops=per_tok * B * steps, bytes=wbytes * steps)
    out["prepare"] = dict(flops=2.0 * (V + B) * E * 4 * H, bytes=4.0 * (V * E + 4 * H * 2 * E + (V + B) * 4 * H))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # r03: 100 / 20 -- the first ~25 batches after start-up run 4-6 % below
    ap.add_argument("--warmup", type=int, default=20)     # the steady state (clock / allocator ramp), 20 / 5 measured that ramp
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--seq", type=int, default=150, help="decode steps (max_length)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipe-rows", type=int, default=2, help="decode rows per workgroup in the pipelined region")
    ap.add_argument("--pipe-decoders", type=int, default=1, help="decode streams in the pipelined region")
    ap.add_argument("--pipe-depth", type=int, default=3,
                    help="batches in flight in the pipelined region (r04: 3 -- with 2 the host enqueues encoder(i+1) only after it has "
                         "collected batch i-1, ~100 us into decode(i); with 3 the encoder's wait kernel is already in its queue)")
    ap.add_argument("--pipe-encoders", type=int, default=None,
                    help="encoder streams in the pipelined region (default 1; 2 for --mode resnet: two trunks side by side "
                         "fill each other's launch gaps and tile tails)")
    ap.add_argument("--resnet-joins", action="store_true",
                    help="--mode resnet: layer1's conv3 + identity + ReLU and the next conv1 in one launch (ResNetEncoder.fuse_joins; A/B)")
    ap.add_argument("--resnet-flags", type=lambda v: int(v, 0), default=0,
                    help="--mode resnet: I2L_FLAG_* bits for the trunk's conv kernels (A/B runs, e.g. 0x60000 = ring kernel "
                         "with 128-column tiles everywhere)")
    ap.add_argument("--mode", choices=["greedy", "beam", "train", "resnet", "preprocess", "metrics"], default="greedy",
                    help="greedy = the headline (BASELINE configs[1]); beam = configs[2] (128 images x k=5, attention); "
                         "train = configs[3] (teacher-forced fwd+bwd+CE+clip+Adam, 64 samples/GPU, RCCL all-reduce); "
                         "resnet = configs[4] (ResNet50 encoder in bf16 + greedy decode, batch 256)")
    ap.add_argument("--no-coresident", action="store_true",
                    help="pipelined region on the r01 scheme (row-per-workgroup decode on half of the CUs) instead of the "
                         "co-resident one: decode of batch i on the 8-member grouped kernel (one wave per SIMD, 80 KB of LDS per "
                         "CU), the conv workgroups of batch i + 1's encoder on the same CUs")
    ap.add_argument("--decode-members", type=int, choices=[8, 16], default=16,
                    help="co-resident pipelined region: the grouped decode kernel -- 8 members x 8 rows per group on the vector ALUs "
                         "(decode_group8_kernel, v_pk_fma_f32) or 16 members x 16 rows on the matrix cores (decode_group16_kernel, "
                         "split-bf16 v_mfma_f32_16x16x32_bf16 bursts)")
    ap.add_argument("--decode-priority", type=int, default=-1, help="pipelined region: HIP stream priority of the decode stream (-1 = high)")
    ap.add_argument("--encoder-priority", type=int, default=0, help="pipelined region: HIP stream priority of the encoder stream(s)")
    ap.add_argument("--pipe-trace-every", type=int, default=8,
                    help="pipelined region: HIP timing events around batches j, j + 1 for every j that is a multiple of this (1 = all)")
    ap.add_argument("--pipe-late-clear", action="store_true",
                    help="pipelined region: the decode launch zeroes its exchange region itself, on the decode stream (until r04 v4; A/B)")
    ap.add_argument("--pipe-no-hold", action="store_true",
                    help="pipelined region, A/B: do NOT hold encoder(i+1) back until decode(i) is resident (the dependency "
                         "GreedyPipeline adds in the co-resident mode: i2l_greedy_decode_ex's residency signal + "
                         "i2l_stream_wait_value32 on the encoder stream)")
    ap.add_argument("--settle-ms", type=float, default=80.0,
                    help="untimed run of the same step for this long BEFORE the W warm-up steps of the first timed region: after "
                         "idle the GPU needs ~50 ms of load to reach its clocks (profiles/r03/ramp.txt: 1.5 -> 1.07 ms per batch "
                         "over the first ~45 batches of a fresh process); 0 = none.  Reported as config.settle, the rate of the "
                         "first K batches WITHOUT it as value_cold_start")
    ap.add_argument("--serial", action="store_true", help="time only batches back to back on one stream (the r02 default)")
    ap.add_argument("--pipelined", action="store_true",
                    help="(default since r03) time the two-stream batch pipeline (GreedyPipeline: encoder of batch i+1 beside the "
                         "decode of batch i); a second, serial pass of the same number of steps provides the undisturbed "
                         "per-kernel times for the roofline object, and `value` is never worse than that serial pass")
    ap.add_argument("--dp-single-allreduce", action="store_true",
                    help="--mode train: ONE all-reduce of the flat gradient buffer after the backward pass instead of the "
                         "default two pieces (same sums), the first of which overlaps the conv backward")
    ap.add_argument("--no-side-wgrad", action="store_true",
                    help="--mode train: keep the weight-gradient kernels on the step's own stream (default: two side lanes owned by TrainStep)")
    ap.add_argument("--launch-probe", action="store_true",
                    help="rendezvous check only (CPU, gloo): every rank all-reduces its rank number, rank 0 prints "
                         "{n_gpus, ranks_seen}; exercises the self-launch path of `--gpus N` without a GPU")
    args = ap.parse_args()
    args.pipelined = not args.serial
    args.coresident = not args.no_coresident
    args.group_flag = _lib.FLAG_DECODE_GROUP16 if args.decode_members == 16 else _lib.FLAG_DECODE_GROUP8
    args.group_kernel = f"decode_group{args.decode_members}_kernel"

    if args.launch_probe:
        os.environ.setdefault("I2L_DIST_BACKEND", "gloo")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves.  Nothing in this process has touched the GPU
        # (GPUs are counted from sysfs), and the ranks are CHILD processes, never an exec of this one
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to report a line whose n_gpus "
                         "is not what was asked for")
    if args.launch_probe:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = torch.tensor([float(rank + 1)])
        dist.all_reduce(seen)
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks_seen": float(seen.item()), "probe": True}), flush=True)
        dist.destroy_process_group()
        return
    local = local % max(1, torch.cuda.device_count())       # rehearsals may put several ranks on one card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("I2L_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.mode != "greedy":
        return extra_modes(args, world, rank, dev, dist)
    B, T = args.batch, args.seq
    cfg = synth.model_config()                                        # primary dims: E=Hd=256, L=1, V=512, 3x64x320
    sd_kw = dict(seed=42, out_scale=8.0, enc_scale=16.0)              # = tests/golden/primary_cfg2 weights
    model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, **sd_kw).items()})
    model = model.to(dev).eval()
    images = torch.from_numpy(synth.make_images(B, cfg, seed=1234 + rank)).to(dev)   # resident in HBM
    ids_host = torch.empty((B, T), dtype=torch.int32).pin_memory()

    events = []                                                       # (name, event) on the launch stream

    def hook(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        events.append((name, ev))

    last = [ids_host]

    def serial_step():
        with torch.no_grad():
            enc = model.encoder(images)
            ids, _ = model.greedy_ids(enc, synth.START, synth.END, T)
            ids_host.copy_(ids, non_blocking=True)

    settled = {"batches": 0, "ms": 0.0}

    def settle(step_fn, drain_fn):
        """Untimed: the same step until --settle-ms of wall time have passed (clock ramp after idle)."""
        if args.settle_ms <= 0:
            return
        t_s = time.perf_counter()
        while (time.perf_counter() - t_s) * 1e3 < args.settle_ms:
            for _ in range(4):
                step_fn()
            drain_fn()
            torch.cuda.synchronize()
            settled["batches"] += 4
        settled["ms"] = round((time.perf_counter() - t_s) * 1e3, 1)

    def timed(step_fn, drain_fn, hooked):
        """W warmup steps, then exactly K steps between barrier + synchronize fences; returns seconds."""
        for _ in range(args.warmup):
            step_fn()
        drain_fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        if hooked:
            _lib.set_stage_hook(hook)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        drain_fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        _lib.set_stage_hook(None)
        return dt

    serial_elapsed = pipelined_elapsed = None
    cold_elapsed = None
    if args.serial:
        settle(serial_step, lambda: None)
        elapsed = timed(serial_step, lambda: None, hooked=True)
        serial_elapsed = elapsed
    else:
        pipe = GreedyPipeline(model, synth.START, synth.END, T, depth=args.pipe_depth,
                              rows_per_workgroup=0 if args.coresident else args.pipe_rows, decode_streams=args.pipe_decoders,
                              decode_flags=args.group_flag if args.coresident else 0,
                              decode_priority=args.decode_priority, encoder_streams=args.pipe_encoders or 1,
                              encoder_priority=args.encoder_priority, hold_encoder=False if args.pipe_no_hold else None, clear_early=not args.pipe_late_clear)

        def pipe_step():
            if pipe.pending() >= pipe.depth:
                last[0] = pipe.collect()
            pipe.submit(images)

        def pipe_drain():
            while pipe.pending():
                last[0] = pipe.collect()

        # straight after process start: W warm-up + K timed batches on clocks that are still ramping (reported, not `value`)
        cold_elapsed = timed(pipe_step, pipe_drain, hooked=False) if args.settle_ms > 0 else None
        settle(pipe_step, pipe_drain)
        # HIP events of the timed region, on the streams the kernels are launched on: the warm-up batches are traced too and
        # dropped below (exactly the last K records are the timed batches)
        pipe.trace = []
        # sampled timing events only when the region is long enough to hold several traced pairs; short runs time every batch
        pipe.trace_every = args.pipe_trace_every if args.steps >= 4 * args.pipe_trace_every else 1
        first_timed = pipe._batch_no + args.warmup
        elapsed = timed(pipe_step, pipe_drain, hooked=False)          # <- the timed region of `value`
        pipe_trace, pipe.trace = [r for r in pipe.trace if r["batch"] >= first_timed], None
        ids_pipe = last[0].clone()
        serial_elapsed = timed(serial_step, lambda: None, hooked=True)  # per-kernel times, undisturbed
        last[0] = ids_host
        # the two regions run different decode kernels (8-member / 4-member groups: fp32 sums in another order), so a row may
        # leave the other region's ids at an fp32 near-tie; BOTH are held against the reference's ids below (margin guard)
        differing = int((ids_pipe != ids_host).any(dim=1).sum())
        assert differing <= 0.02 * B, f"pipelined and serial search disagree on {differing} rows"
        pipelined_elapsed = elapsed
        if serial_elapsed < elapsed:       # stream overlap is up to the hardware queues: never report worse than serial
            elapsed = serial_elapsed

    # steps actually executed by the reference's stop rule (all rows END in one step, seq2seq.py:220)
    ids_host = last[0]
    all_end = (ids_host == synth.END).all(dim=0)
    executed = int(all_end.nonzero()[0]) + 1 if bool(all_end.any()) else T
    tokens_per_step = B * executed
    executed_steps_holder = [executed]

    if dist is not None:
        tmax = torch.tensor([elapsed, serial_elapsed, pipelined_elapsed or 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        serial_elapsed = float(tmax[1].item())
        if pipelined_elapsed is not None:
            pipelined_elapsed = float(tmax[2].item())
            elapsed = min(serial_elapsed, pipelined_elapsed)
        else:
            elapsed = serial_elapsed
        tok = torch.tensor([tokens_per_step], dtype=torch.float64, device=dev)
        dist.all_reduce(tok, op=dist.ReduceOp.SUM)
        total_tokens_per_step = float(tok.item())
    else:
        total_tokens_per_step = float(tokens_per_step)

    # per-stage device time from the HIP events recorded inside the timed region
    stage_ms = {}
    for (n0, e0), (n1, e1) in zip(events[:-1], events[1:]):
        if n1 == "begin":
            continue
        stage_ms.setdefault(n1, []).append(e0.elapsed_time(e1))
    stage_avg = {k: float(np.mean(v)) for k, v in stage_ms.items()}
    costs = stage_costs(cfg, B, executed)
    stages = []
    for name, ms in stage_avg.items():
        c = costs.get(name)
        if c is None:
            continue
        stages.append(dict(kernel=name, ms=round(ms, 4), tflops=round(c["flops"] / ms / 1e9, 3),
                           gbs=round(c["bytes"] / ms / 1e6, 2),
                           frac_fp32=round(c["flops"] / ms / 1e9 / PEAK_FP32_TFLOPS, 4),
                           frac_hbm=round(c["bytes"] / ms / 1e6 / PEAK_HBM_GBS, 4)))
    dom = max(stages, key=lambda s: s["ms"])
    # HBM bytes per launch of the dominant kernel: hardware counters cannot be read inside this run (rocprofv3 --pmc
    # is its own pass and serialises the kernels), so `traffic` is the figure of the LATEST committed counter collection
    # of this same command (profiles/collect.sh -> profiles/traffic.json), and `traffic_from` says which one
    tj, traffic_from = {}, None
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        traffic_from = tj.get("_collection", "profiles/traffic.json (rocprofv3 --pmc passes of `python bench.py`, collection "
                                             "not recorded)") + " -- a separate counter pass, not this timed run"
    H_, V_, L_ = cfg["hidden_dim"], cfg["vocab_size"], cfg["lstm_layers"]
    run_flops = 2.0 * (4 * H_ * H_ * (2 * L_ - 1) + H_ * V_) * B * executed_steps_holder[0]

    def decode_view(kernel, ms, members):
        """Roofline entry of a grouped decode kernel.  The loop's matrix work (1.835 MFLOP/token algorithmic) is
        priced against the fp32 peak as the contract asks, but the kernel is NOT arithmetic-bound: it runs on the vector ALUs
        (v_pk_fma_f32, no MFMA), executes only 0.786 MFLOP/token (the W_ih [emb | enc] half of the gate GEMM is hoisted into
        the P / Genc tables by `prepare`), and every one of its steps waits on two exchanges between the workgroups of a
        group through L2: a latency chain."""
        tf = costs["decode"]["flops"] / ms / 1e9
        on_mfma = members == 16
        return dict(kernel=kernel, bound="latency", achieved=round(tf, 3), peak=PEAK_FP32_TFLOPS, unit="TFLOP/s",
                    frac=round(tf / PEAK_FP32_TFLOPS, 4), launch_ms=round(ms, 4),
                    traffic=tj.get({4: "decode", 8: "decode8", 16: "decode16"}[members]), traffic_from=traffic_from,
                    # matrix-pipe busy fraction of this kernel (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x clock)) from the
                    # same committed counter collection: north_star asks for >= 0.5 on the LSTM GEMMs -- a latency chain of 150
                    # steps with 72 MFMAs per wave and step cannot fill the pipe (0 for the kernels on the vector ALUs)
                    mfma_busy=tj.get("_mfma_busy", {}).get({4: "decode", 8: "decode8", 16: "decode16"}[members]),
                    priced_against="mfma (fp32 peak); the kernel's products run " + ("as split-bf16 v_mfma_f32_16x16x32_bf16 (6 bf16 "
                    "partial products per fp32 product: 4.7 MFLOP/token executed on the bf16 pipe)" if on_mfma else
                    "as v_pk_fma_f32 on the vector ALUs, same 157.3 TFLOP/s peak"),
                    executed_tflops=round(run_flops / ms / 1e9, 3),
                    frac_executed=round(run_flops / ms / 1e9 / PEAK_FP32_TFLOPS, 4),
                    us_per_decode_step=round(ms * 1e3 / executed_steps_holder[0], 3),
                    note=f"per-step latency chain (two in-group L2 exchanges among {members} members + LSTM cell), not arithmetic: "
                         "frac = algorithmic 1.835 MFLOP/token, frac_executed = the 0.786 MFLOP/token the kernel runs")

    # ---- the serial pass's dominant kernel (one stream, every kernel alone on the chip)
    if dom["kernel"] == "decode":
        serial_view = decode_view("decode_group_kernel", dom["ms"], 4)
        gst = model.decoder.group_status()
        if gst is not None:     # measured placement of the last launch: L2-local exchange needs a group's 4 workgroups on one XCD
            serial_view["groups_on_one_xcd"] = f"{gst['groups_on_one_xcd']}/{gst['groups']}"
    else:
        serial_view = dict(kernel=dom["kernel"], bound="mfma", achieved=dom["tflops"], peak=PEAK_FP32_TFLOPS, unit="TFLOP/s",
                           frac=round(dom["tflops"] / PEAK_FP32_TFLOPS, 4), launch_ms=dom["ms"], traffic=tj.get(dom["kernel"]),
                           traffic_from=traffic_from)
    serial_view["measured_in"] = (f"{'the timed region' if args.serial else 'a serial pass of ' + str(args.steps) + ' steps right after the timed region'}"
                                  " (one stream, whole chip per kernel; stage marks = HIP events on that stream)")
    schedule = None
    timed_is_pipeline = (not args.serial) and args.coresident and pipelined_elapsed is not None and elapsed == pipelined_elapsed
    if (not args.serial) and args.coresident:
        # ---- the TIMED region's dominant kernel (decode_group16_kernel / decode_group8_kernel), from the HIP events GreedyPipeline recorded on the decode
        # stream around each of the K launches of the timed region (the launch = a memset node of the exchange region + the
        # kernel); beside it the encoder chain of the same region and the schedule the two streams were in
        dec_ms = np.array([r["dec_start"].elapsed_time(r["dec_end"]) for r in pipe_trace])
        enc_ms = np.array([r["enc_start"].elapsed_time(r["enc_end"]) for r in pipe_trace])
        # encoder(i + 1) must start AFTER decode(i) is on the compute units; offset = its first kernel's start behind the decode launch
        pairs = [(a, b) for a, b in zip(pipe_trace[:-1], pipe_trace[1:]) if b["batch"] == a["batch"] + 1]
        offs = np.array([a["dec_start"].elapsed_time(b["enc_start"]) for a, b in pairs])
        # ... and in the fast schedule the whole encoder(i + 1) ends before decode(i) does (profiles/r03/ramp.txt)
        inside = np.array([b["enc_end"].elapsed_time(a["dec_end"]) for a, b in pairs])
        with torch.no_grad():     # the same kernel alone on the chip
            enc_ = model.encoder(images)
            for _ in range(3):
                model.greedy_ids(enc_, synth.START, synth.END, T, flags=args.group_flag)
            prep = model.decoder.prepare(enc_)
            prepared = (prep[0], prep[1], prep[2], model.decoder._ws)
            ev8 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev8[0].record()
            for _ in range(10):
                model.greedy_ids(enc_, synth.START, synth.END, T, flags=args.group_flag, prepared=prepared)
            ev8[1].record()
            torch.cuda.synchronize()
        alone_ms = ev8[0].elapsed_time(ev8[1]) / 10
        pipe_view = decode_view(args.group_kernel, float(dec_ms.mean()), args.decode_members)
        pipe_view.update(launch_ms_min_max=[round(float(dec_ms.min()), 4), round(float(dec_ms.max()), 4)],
                         alone_ms=round(alone_ms, 4), frac_alone=round(costs["decode"]["flops"] / alone_ms / 1e9 / PEAK_FP32_TFLOPS, 4),
                         stretch_beside_encoder=round(float(dec_ms.mean()) / alone_ms, 3),
                         encoder_chain_ms_in_region=round(float(enc_ms.mean()), 4),
                         measured_in=f"the timed region: HIP events on the decode stream around {len(dec_ms)} of its {args.steps} launches "
                                     f"(batches j and j + 1 for every j that is a multiple of {pipe.trace_every}: four timing events per batch "
                                     "cost the pipeline ~0.7 %; --pipe-trace-every 1 times every launch) "
                                     "(alone_ms: the same kernel by itself on the chip, after the region)")
        gst = model.decoder.group_status()
        if gst is not None:
            pipe_view["groups_on_one_xcd"] = f"{gst['groups_on_one_xcd']}/{gst['groups']}"
        schedule = dict(
            held=bool(pipe.hold_encoder),
            encoder_after_decode_launch=round(float((offs > 0).mean()), 4) if len(offs) else None,
            encoder_inside_decode=round(float((inside > 0).mean()), 4) if len(inside) else None,
            encoder_start_behind_decode_launch_us=[round(float(np.percentile(offs, q)) * 1e3, 1) for q in (5, 50, 95)] if len(offs) else None,
            note="fractions of the timed batches: encoder(i+1)'s first kernel started after decode(i) was launched / the whole "
                 "encoder(i+1) ended before decode(i) did -- 1.0 / ~1.0 is the fast schedule (conv workgroups beside a resident "
                 "decode), anything else the slow one (profiles/r03/ramp.txt); percentiles 5 / 50 / 95 of the start offset")
    if timed_is_pipeline:
        roofline = pipe_view
        roofline["serial"] = serial_view
    else:
        roofline = serial_view
        if (not args.serial) and args.coresident:
            roofline["pipelined"] = pipe_view
    enc_stages = [st for st in stages if st["kernel"].startswith("conv") or st["kernel"] == "fc"]
    if enc_stages:
        e_ms = sum(st["ms"] for st in enc_stages)
        e_flops = sum(costs[st["kernel"]]["flops"] for st in enc_stages)
        e_bytes = sum(costs[st["kernel"]]["bytes"] for st in enc_stages)
        roofline["encoder"] = dict(ms=round(e_ms, 4), frac_hbm=round(e_bytes / e_ms / 1e6 / PEAK_HBM_GBS, 4),
                                   frac_fp32=round(e_flops / e_ms / 1e9 / PEAK_FP32_TFLOPS, 4),
                                   frac_bf16_executed=round(6.0 * e_flops / e_ms / 1e9 / PEAK_BF16_TFLOPS, 4),
                                   frac_bf16_executed_of_sustained=round(6.0 * e_flops / e_ms / 1e9 / SUSTAINED_BF16_TFLOPS, 4),
                                   measured_in="the serial pass (each kernel alone on the chip)",
                                   note="algorithmic bytes / flops (SURVEY 8d); executed = 6 bf16 partial products per fp32 product; sustained = "
                                        "what bare MFMA loops reach on random operands (1.82 of the nominal 2.5 PFLOP/s)")
    roofline["stages"] = stages
    roofline["stages_measured_in"] = serial_view["measured_in"]

    result = {
        "metric": "decoded LaTeX tokens/sec at batch 256, 320x64 imgs, seq 150",
        "value": round(total_tokens_per_step * args.steps / elapsed, 1),
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "cnn_lstm greedy decode (BASELINE configs[1])", "batch_per_gpu": B,
                   "global_batch": B * world, "image": "3x64x320", "decode_steps": executed,
                   "embedding_dim": cfg["embedding_dim"], "hidden_dim": cfg["hidden_dim"],
                   "lstm_layers": cfg["lstm_layers"], "vocab": cfg["vocab_size"],
                   "parallelism": f"replicas x{world} (no collective)",
                   "batch_pipeline": "serial" if (args.serial or elapsed == serial_elapsed and pipelined_elapsed != serial_elapsed)
                   else (f"2 streams, co-resident: decode(i) on {args.group_kernel} ({args.decode_members} members x {args.decode_members} rows per "
                         f"group, one wave per SIMD) and the conv workgroups of encoder(i+1) share every CU; {args.pipe_depth} batches in flight"
                         if args.coresident else
                         f"{1 + args.pipe_decoders} streams: encoder(i+1) beside decode(i), {args.pipe_rows} rows/workgroup, "
                         f"{args.pipe_depth} batches in flight")},
        "roofline": roofline,
        "schedule": schedule,
        "value_serial": round(total_tokens_per_step * args.steps / serial_elapsed, 1),
        "value_pipelined": None if pipelined_elapsed is None else round(total_tokens_per_step * args.steps / pipelined_elapsed, 1),
        # the same W + K batches run FIRST, straight after process start, while the GPU's clocks are still ramping up from
        # idle: what a 25 ms run measures without the settle phase (per-rank figure x ranks; not max-reduced)
        "value_cold_start": None if cold_elapsed is None else round(total_tokens_per_step * args.steps / cold_elapsed, 1),
    }
    # what ran before the K timed batches of `value`, in batches: the cold region (W + K), the settle phase, the W warm-up steps
    result["config"]["warmup_effective"] = (args.warmup + (args.warmup + args.steps if cold_elapsed is not None else 0)
                                            + settled["batches"])
    result["config"]["settle"] = (f"{settled['batches']} untimed batches ({settled['ms']} ms) before the {args.warmup} warm-up steps of "
                                  "the timed region: after idle the GPU needs ~50 ms of load to reach its clocks "
                                  "(profiles/r03/ramp.txt); --settle-ms 0 turns it off") if settled["batches"] else "none"

    # ---- outside the timed region: the decode must not have timed out, and the ids must be the reference's
    _lib.check_ids(ids_host)
    result["ids_check"] = golden_ids_check(ids_host, B, T, rank)
    if not args.serial:
        _lib.check_ids(ids_pipe)
        result["ids_check_pipelined"] = golden_ids_check(ids_pipe, B, T, rank)
    # ---- beside the kernel-only headline (NOT `value`): the same batch through the reference-shaped entry points
    if rank == 0 and world == 1:
        result["beside"] = beside_paths(model, cfg, images, B, T, max(3, min(10, args.steps)))
    if rank == 0:
        print(f"[bench] gpu: {result['value']:.0f} tokens/s, {result['ms_per_step']:.3f} ms/step", file=sys.stderr,
              flush=True)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg, sd_kw, B, T)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


def visible_gpus():
    """GPUs this process may use, WITHOUT touching the HIP runtime in the launching parent: KFD topology nodes with SIMDs
    (CPU nodes have simd_count 0), cut down by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set."""
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for d in os.listdir(root):
            try:
                props = dict(l.split() for l in open(os.path.join(root, d, "properties")) if len(l.split()) == 2)
                n += 1 if int(props.get("simd_count", "0")) > 0 else 0
            except OSError:
                pass
    except OSError:
        return 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: N child processes of this script, one rank per GPU, with the
    rendezvous variables torch.distributed.run would set (127.0.0.1, a free port).  Rank 0's stdout (the ONE JSON line)
    is passed through; the exit code is the worst of the ranks'.  Refuses when fewer than N GPUs are visible -- unless
    I2L_DIST_BACKEND=gloo, the one-card rehearsal in which several ranks share a device."""
    import socket
    import subprocess
    have = visible_gpus()
    if have < n and os.environ.get("I2L_DIST_BACKEND", "nccl") == "nccl":
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible; not running a smaller job under that name",
              file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [p.wait() for p in procs]
    return max((abs(c) for c in codes), default=0)


def golden_ids_check(ids_host, B, T, rank):
    """Bench ids against tests/golden/primary_cfg2.npz (ids + top1-top2 margins of the REAL reference at B=256 x 150):
    equal up to each row's first step whose reference margin is below 2e-4 (an fp32 near-tie).  Only the default
    workload on rank 0's images has a fixture."""
    path = os.path.join(REPO, "tests", "golden", "primary_cfg2.npz")
    if rank != 0 or B != 256 or T != 150 or not os.path.exists(path):
        return "skipped (no fixture for this batch / rank)"
    d = np.load(path)
    ref, margins = d["ids"].astype(np.int64)[:, 1:], d["margins"]
    got = ids_host.numpy().astype(np.int64)[:, : ref.shape[1]]
    diverged = 0
    for r in range(B):
        ne = np.nonzero(got[r] != ref[r])[0]
        if ne.size:
            assert margins[r, int(ne[0])] < 2e-4, f"bench ids differ from the reference at row {r} step {int(ne[0])}"
            diverged += 1
    assert diverged <= 0.05 * B, f"{diverged} rows diverge from the reference ids"
    return f"ids == reference fixture primary_cfg2 ({B}x{ref.shape[1]}; {diverged} rows leave it at a near-tie)"


def beside_paths(model, cfg, images, B, T, reps):
    """Two boundary-inclusive measurements next to the headline (same weights, same batch, `reps` passes each):
    (1) Seq2SeqModel.inference() as the reference's callers use it (seq2seq.py:124-190): encoder + search + the
        device->host copy + the Python lists it returns;
    (2) the evaluate chain of cli.py:449-495 as one device chain (Predictor.evaluate_batch): decoded uint8 pages on
        the host -> plan + upload + preprocess kernels -> encoder -> sticky greedy loop -> id compaction ->
        BLEU-4 / Levenshtein statistics -> float64 scores."""
    from img2latex_amd.training import Predictor, TokenTable
    out = {}
    with torch.no_grad():
        model.inference(images, synth.START, synth.END, max_length=T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            seqs = model.inference(images, synth.START, synth.END, max_length=T)
        dt = (time.perf_counter() - t0) / reps
    toks = sum(len(r) - 1 for r in seqs)
    out["inference_api"] = {"tokens_per_s": round(toks / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
                            "includes": "encoder, greedy search, ids device->host, List[List[int]] building"}
    out["host_images"] = host_image_path(model, images, B, T, reps)
    vocab = {"<PAD>": 0, "<START>": 1, "<END>": 2, "<UNK>": 3}
    vocab.update({f"t{i}": i for i in range(4, cfg["vocab_size"])})
    pred = Predictor(model, TokenTable(vocab, max_sequence_length=T), device=images.device)
    sizes = [(30 + (7 * k) % 90, 80 + (53 * k) % 700, 1 + 2 * (k % 2)) for k in range(B)]
    pages = []
    for k, (h, w, c) in enumerate(sizes):
        base = synth.uniform(5000 + k, "img", (h, w, c), 0.0, 255.0)
        pages.append(np.round(base).astype(np.uint8).reshape((h, w) if c == 1 else (h, w, 3)))
    targets = torch.from_numpy(synth.make_formulas(B, T, cfg["vocab_size"], seed=777))
    tg_dev = targets.to(images.device)
    pred.evaluate_batch(pages, tg_dev, max_length=T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = pred.evaluate_batch(pages, tg_dev, max_length=T)
    dt = (time.perf_counter() - t0) / reps
    list(pred.evaluate_stream([(pages, tg_dev)] * 3, max_length=T))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_stream = max(reps, 40)          # three batches in flight: 10 batches would mostly measure the pipeline's fill and drain
    res_s = list(pred.evaluate_stream([(pages, tg_dev)] * n_stream, max_length=T))
    dt_stream = (time.perf_counter() - t0) / n_stream
    assert res_s[-1]["bleu"] == res["bleu"] and res_s[-1]["levenshtein"] == res["levenshtein"]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        res2 = pred.evaluate_batch(images, tg_dev, max_length=T)
    ev[1].record()
    torch.cuda.synchronize()
    # the stream from tensors already on the device: what the device-side schedule of evaluate_stream alone is worth
    list(pred.evaluate_stream([(images, tg_dev)] * 3, max_length=T))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_dev = max(4 * reps, 40)
    res_d = list(pred.evaluate_stream([(images, tg_dev)] * n_dev, max_length=T))
    dt_dev_stream = (time.perf_counter() - t0) / n_dev
    assert res_d[-1]["bleu"] == res2["bleu"] and abs(res_d[-1]["levenshtein"] - res2["levenshtein"]) <= 1e-3
    out["evaluate_chain"] = {"tokens_per_s": round(B * T / dt, 1), "images_per_s": round(B / dt, 1),
                             "ms_per_batch_stream_from_device_tensors": round(dt_dev_stream * 1e3, 3),
                             "ms_per_batch": round(dt * 1e3, 3),
                             "ms_per_batch_pipelined": round(dt_stream * 1e3, 3),
                             "ms_per_batch_from_device_tensors": round(ev[0].elapsed_time(ev[1]) / reps, 3),
                             "bleu": res["bleu"], "levenshtein": res["levenshtein"],
                             "pipelined": "Predictor.evaluate_stream: host packing + upload (side stream) of batch i+1 beside the kernels of batch i; on the device decode(i) beside preprocessing + encoder(i+1) (two streams, as GreedyPipeline); from ragged host pages the host is the bound, ms_per_batch_stream_from_device_tensors shows the device side alone",
                             "includes": "host resize plans + upload of ragged uint8 pages, preprocess, encoder, decode "
                                         f"({T} steps, sticky stop), id compaction, sequence statistics, float64 scores"}
    return out


def host_image_path(model, images, B, T, reps):
    """The PCIe-inclusive rate of the headline workload: the fp32 pixels start in (pinned) host memory every batch, as
    they do behind the reference's DataLoader (dataset.py:456-464).  The upload of batch i+1 runs on a copy stream into
    the second of two device buffers while batch i is encoded and decoded; the ids go back to pinned host memory on a
    third stream (on the compute stream that small copy would queue behind the next 63 MB upload in the DMA engine and
    hold the next encoder back: 2.1 instead of 1.4 ms per batch)."""
    dev = images.device
    host = images.cpu().pin_memory()
    bufs = [torch.empty_like(images) for _ in range(2)]
    ids_host = [torch.empty(B, T, dtype=torch.int32).pin_memory() for _ in range(2)]
    up_s, down_s, main_s = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.current_stream(dev)
    uploaded = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]

    def upload(i):
        with torch.cuda.stream(up_s):
            up_s.wait_event(consumed[i & 1])
            bufs[i & 1].copy_(host, non_blocking=True)
            uploaded[i & 1].record(up_s)

    def run(n):
        for e in consumed:
            e.record(main_s)
        upload(0)
        for i in range(n):
            if i + 1 < n:
                upload(i + 1)
            main_s.wait_event(uploaded[i & 1])
            enc = model.encoder(bufs[i & 1])
            ids, _ = model.greedy_ids(enc, synth.START, synth.END, T)
            consumed[i & 1].record(main_s)
            with torch.cuda.stream(down_s):
                down_s.wait_event(consumed[i & 1])
                ids_host[i & 1].copy_(ids, non_blocking=True)
                ids.record_stream(down_s)
        torch.cuda.synchronize()

    with torch.no_grad():
        run(30)                 # the runtime's copy queues and signal pools settle over the first ~20 batches (2-4 ms hiccups)
        n = max(reps, 40)
        t0 = time.perf_counter()
        run(n)
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(5):
            bufs[0].copy_(host, non_blocking=True)
        torch.cuda.synchronize()
        up = (time.perf_counter() - t0) / 5
    for h in ids_host:
        _lib.check_ids(h)
    mb = host.numel() * 4 / 1e6
    return {"tokens_per_s": round(B * T / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
            "upload_alone_ms": round(up * 1e3, 3), "upload_gb_per_s": round(mb / up / 1e3, 1),
            "includes": f"{mb:.0f} MB of fp32 pixels from pinned host memory per batch (two device buffers, upload stream), "
                        "encoder, greedy search, ids device->host (download stream)"}


def extra_modes(args, world, rank, dev, dist):
    """Secondary configs of BASELINE.json (not the driver's headline line)."""
    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.mode == "beam":
        n, k, T = 128, 5, args.seq
        cfg = synth.model_config(attention=True)
        sd_kw = dict(seed=42, out_scale=12.0, enc_scale=16.0, end_clock=(0.05, 12.0, 6.0))
        model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
        model.load_state_dict({k_: torch.from_numpy(v) for k_, v in synth.make_state_dict(cfg, **sd_kw).items()})
        model = model.to(dev).eval()
        images = torch.from_numpy(synth.make_images(n, cfg, seed=1234 + rank)).to(dev)
        out = [None]

        def one_step():
            with torch.no_grad():
                out[0] = model.beam_search_batch(model.encoder(images), synth.START, synth.END, T, k)
        unit = lambda: float(sum(len(s) for s in out[0]))

        def roof(ms):       # beam-steps actually needed by the returned sequences x the decoder's algorithmic flops
            steps = sum(min(len(s) + 1, T) for s in out[0])
            fl = 1.835e6 * k * steps
            return dict(bound="latency", kernel="beam_group_kernel<5>", achieved=round(fl / ms / 1e9, 3),
                        peak=PEAK_FP32_TFLOPS, unit="TFLOP/s", frac=round(fl / ms / 1e9 / PEAK_FP32_TFLOPS, 4), traffic=None,
                        note="algorithmic 1.835 MFLOP per beam-step over sum(len+1) x k beam-steps; per-step latency "
                             "chain of in-group exchanges like the greedy kernel, vector ALUs")

        def cpu_base():
            import img2latex_oracle as O
            sd = O.to_torch_sd(synth.make_state_dict(cfg, **sd_kw))
            x = torch.from_numpy(synth.make_images(n, cfg, seed=1234)[:4])
            with torch.no_grad():
                enc = O.cnn_encoder(sd, cfg, x)
                t0 = time.perf_counter()
                seqs = [O.beam_search(sd, cfg, enc[j:j + 1], synth.START, synth.END, T, k) for j in range(4)]
                dt = time.perf_counter() - t0
            return {"value": round(sum(len(q) for q in seqs) / dt, 1), "unit": "tokens/s", "cores": torch.get_num_threads(),
                    "kind": "port", "sample": f"4 of the {n} images, k={k}, beam search only (encoder excluded)"}
        name, conf = "beam-search decoded tokens/sec (best sequences)", {"workload": "cnn_lstm+attention beam k=5 (BASELINE configs[2])", "images_per_gpu": n, "beam": k, "max_length": T}

        def after_check():
            """Outside the timed region: the sequences are the REFERENCE's (tests/golden/primary_cfg3_beam.npz: 128 images,
            k = 5, max_length 150 -- tokens + per-image decision margins read from the reference's own search)."""
            path = os.path.join(REPO, "tests", "golden", "primary_cfg3_beam.npz")
            if rank != 0 or T != 150 or not os.path.exists(path):
                return "skipped (no fixture for this rank / length)"
            d = np.load(path)
            off = 0
            for j in range(n):
                want = [int(t) for t in d["ids"][j, :d["lens"][j]]]
                if out[0][j] != want:
                    assert d["min_gap"][j] < 1e-4, f"beam tokens of image {j} differ from the reference's"
                    off += 1
            assert off <= 0.05 * n
            return f"tokens == reference fixture primary_cfg3_beam ({n} images; {off} leave it at a near-tie)"
    elif args.mode == "resnet":
        Bn, T = args.batch, args.seq
        cfg = synth.model_config()
        enc_p = dict(img_height=64, img_width=320, channels=3, model_name="resnet50", embedding_dim=256,
                     freeze_backbone=True)
        model = Seq2SeqModel("resnet_lstm", cfg["vocab_size"], enc_p, synth.decoder_params(cfg))
        shapes = [(k_, tuple(v.shape)) for k_, v in model.encoder.state_dict().items()]
        esd = synth.make_resnet_state_dict(shapes, seed=5)
        dsd = {k_: v for k_, v in synth.make_state_dict(cfg, seed=42, out_scale=8.0).items() if k_.startswith("decoder.")}
        full = {"encoder." + k_: torch.from_numpy(v) for k_, v in esd.items()}
        full.update({k_: torch.from_numpy(v) for k_, v in dsd.items()})
        model.load_state_dict(full)
        model = model.to(dev).eval()
        model.encoder.kernel_flags = args.resnet_flags
        model.encoder.fuse_joins = args.resnet_joins
        images = torch.from_numpy(synth.make_images(Bn, cfg, seed=1234 + rank)).to(dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        enc_ms = []
        serial_ids = [None]

        def one_step():
            with torch.no_grad():
                ev[0].record()
                enc = model.encoder(images)
                ev[1].record()
                ids, _ = model.greedy_ids(enc, synth.START, synth.END, T)
                serial_ids[0] = ids.cpu()
                enc_ms.append(ev[0].elapsed_time(ev[1]))
        unit = lambda: float(Bn * T)
        # ResNet-50 trunk at 64x320: 4.09 GMAC x (64*320)/(224*224)
        gflop = 2 * 4.09 * (64 * 320) / (224 * 224) * Bn

        def layer_bytes():
            """Algorithmic HBM bytes of the trunk AS A CHAIN OF LAYER KERNELS (every activation written once by its producer
            and read once by each consumer, bf16; filters once): what bounds it at this batch and image size."""
            model.encoder.trace = []
            with torch.no_grad():
                model.encoder(images)
            torch.cuda.synchronize()
            tot = 0.0
            for conv, bn, x_, res_, y_, relu_, nchw in model.encoder.trace:
                need = x_.numel() / (4 if (conv.kernel_size[0] == 1 and conv.stride[0] == 2) else 1)   # a strided 1x1 reads a quarter
                tot += need * (4 if nchw else 2) + y_.numel() * 2 + conv.weight.numel() * 2
                tot += res_.numel() * 2 if res_ is not None else 0
            stem_out = model.encoder.trace[0][4].numel() * 2
            tot += stem_out + stem_out / 4                       # 3x3 / 2 max-pool: reads the stem's output, writes a quarter
            tot += model.encoder.trace[-1][4].numel() * 2        # global average pool
            model.encoder.trace = None
            return tot

        def roof(ms):
            e_ms = float(np.median(enc_ms[-args.steps:]))
            byts = layer_bytes()
            return dict(bound="mfma", kernel="gemm_bf16_ring_kernel chain (53 conv launches of the trunk)",
                        achieved=round(gflop / e_ms, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                        frac=round(gflop / e_ms / PEAK_BF16_TFLOPS, 4), traffic=None, launch_ms=round(e_ms, 4),
                        note="whole encoder (stem, max-pool, 52 GEMM-shaped convs, average pool, FC) against the dense bf16 peak "
                             "(SURVEY 8d cfg5); see hbm_view: layer by layer the trunk at this size is nearer the HBM roofline",
                        hbm_view=dict(algorithmic_bytes=round(byts), flop_per_byte=round(gflop * 1e9 / byts, 1),
                                      ridge_flop_per_byte=round(PEAK_BF16_TFLOPS * 1e3 / PEAK_HBM_GBS, 1),
                                      achieved_gbs=round(byts / e_ms / 1e6, 1), peak_gbs=PEAK_HBM_GBS,
                                      frac=round(byts / e_ms / 1e6 / PEAK_HBM_GBS, 4),
                                      note="every activation written once and read once per consumer (bf16), filters once: "
                                           "below the ridge, i.e. as a chain of layer kernels the trunk is HBM-bound at "
                                           "B=256 x 64x320; only cross-layer fusion moves this bound"))

        def cpu_base():
            import img2latex_oracle as O
            import resnet_oracle as RO
            sdt = {k_: torch.from_numpy(v) for k_, v in dsd.items()}
            sdt.update({"encoder." + k_: torch.from_numpy(v) for k_, v in esd.items()})
            x = torch.from_numpy(synth.make_images(Bn, cfg, seed=1234)[:16])
            with torch.no_grad():
                t0 = time.perf_counter()
                enc = RO.resnet_encoder(sdt, "resnet50", x)
                ids = O.greedy_search(sdt, cfg, enc, synth.START, synth.END, T)
                dt = time.perf_counter() - t0
            return {"value": round(16 * (len(ids[0]) - 1) / dt, 1), "unit": "tokens/s", "cores": torch.get_num_threads(),
                    "kind": "port", "sample": f"16 of the {Bn} images, fp32 ResNet-50 restatement + {T} greedy steps"}
        name = "decoded LaTeX tokens/sec, ResNet50 bf16 encoder + LSTM greedy decode"
        conf = {"workload": "resnet50_lstm greedy (BASELINE configs[4])", "batch_per_gpu": Bn, "image": "3x64x320",
                "decode_steps": T, "encoder_dtype": "bf16 (fp32 accumulate)", "encoder_ms": enc_ms,
                "encoder_gflop": round(gflop, 1)}
    elif args.mode == "preprocess":
        # load_image after decoding for a ragged batch of page-like uint8 images -> (B, 3, 64, 320)
        from img2latex_amd.data import preprocess_batch
        Bn = args.batch
        sizes = [(30 + (7 * k) % 90, 80 + (53 * k) % 700, 1 + 2 * (k % 2)) for k in range(Bn)]
        imgs = []
        for k, (h, w, c) in enumerate(sizes):
            base = synth.uniform(5000 + k + 1000 * rank, "img", (h, w, c), 0.0, 255.0)
            imgs.append(np.round(base).astype(np.uint8).reshape((h, w) if c == 1 else (h, w, 3)))
        out = [None]

        def one_step():
            out[0] = preprocess_batch(imgs, (64, 320), 3, True)
        unit = lambda: float(Bn)
        in_bytes = float(sum(a.size for a in imgs))

        def roof(ms):       # algorithmic bytes: every source pixel read once, every output float written once
            byts = in_bytes + 4.0 * Bn * 3 * 64 * 320
            return dict(bound="hbm", kernel="resize_h_kernel + resize_v_norm_kernel (+ host plan and upload in the step)",
                        achieved=round(byts / ms / 1e6, 2), peak=PEAK_HBM_GBS, unit="GB/s",
                        frac=round(byts / ms / 1e6 / PEAK_HBM_GBS, 5), traffic=None,
                        note="the step is dominated by host-side planning and the PCIe upload of the pixels, not by the kernels")
        cpu_base = None
        name = "preprocessed images/sec (convert + LANCZOS resize + pad/crop + normalise, host plan + upload included)"
        conf = {"workload": "load_image tail for a ragged batch (SURVEY 8f-3)", "batch_per_gpu": Bn, "out": "3x64x320",
                "unit": "images/s"}
        if rank == 0:
            from PIL import Image
            t0 = time.perf_counter()
            for a in imgs:                                  # the reference's CPU path: Pillow resize + numpy/torch tail
                im = Image.fromarray(a, "L" if a.ndim == 2 else "RGB").convert("RGB")
                nw = int(round(64 * (im.size[0] / im.size[1])))
                im = im.resize((nw, 64), Image.Resampling.LANCZOS)
                canvas = Image.new("RGB", (320, 64), 255)
                canvas.paste(im.crop(((nw - 320) // 2, 0, (nw - 320) // 2 + 320, 64)) if nw > 320 else im, (0, 0))
                t = torch.from_numpy(np.transpose(np.array(canvas), (2, 0, 1))).float() / 255.0
                t = (t - torch.tensor([0.485, 0.456, 0.406]).view(-1, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(-1, 1, 1)
            pil_base = {"value": round(Bn / (time.perf_counter() - t0), 1), "unit": "images/s", "cores": 1,
                        "kind": "reference", "sample": f"{Bn} images through Pillow's resize + the numpy / torch tail of "
                                                       "load_image (the reference's own dependency, one thread)"}
            cpu_base = lambda: pil_base
    elif args.mode == "metrics":
        # calculate_metrics (BLEU-4 + Levenshtein) for a batch of decoded sequences
        from img2latex_amd.training import metrics as M
        Bn, T = args.batch, args.seq
        preds = [synth.randint(900 + k, "p", (T - (k % 17),), 4, 64).tolist() for k in range(Bn)]
        tgts = [synth.randint(901 + k, "t", (T - (k % 11),), 4, 64).tolist() for k in range(Bn)]
        out = [None]

        def one_step():
            out[0] = M.calculate_metrics(preds, tgts)
        unit = lambda: float(Bn)

        def roof(ms):       # integer DP in LDS: the only HBM traffic is the ids in and 7 ints per pair out
            byts = 4.0 * (sum(len(q) for q in preds) + sum(len(q) for q in tgts) + 9 * Bn)
            return dict(bound="hbm", kernel="sequence_metrics_kernel", achieved=round(byts / ms / 1e6, 3), peak=PEAK_HBM_GBS,
                        unit="GB/s", frac=round(byts / ms / 1e6 / PEAK_HBM_GBS, 6), traffic=None,
                        note="LDS-resident anti-diagonal DP, one workgroup per pair: latency of T dependent diagonals, "
                             "HBM is idle; the step also packs and uploads the Python lists")
        cpu_base = None
        name = "sequence pairs/sec (BLEU-4 + Levenshtein, list packing + upload included)"
        conf = {"workload": "calculate_metrics (SURVEY 8f-4)", "pairs": Bn, "tokens_per_sequence": T, "unit": "pairs/s"}
        if rank == 0:
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            import metrics_oracle as MO
            t0 = time.perf_counter()
            ref = MO.calculate_metrics(preds[:32], tgts[:32])
            m_base = {"value": round(32 / (time.perf_counter() - t0), 1), "unit": "pairs/s", "cores": 1,
                      "kind": "port", "sample": "32 of the pairs"}
            cpu_base = lambda: m_base
    else:
        Bt, T = 64, 150
        cfg = synth.model_config(dropout=0.1)
        model = Seq2SeqModel("cnn_lstm", cfg["vocab_size"], synth.encoder_params(cfg), synth.decoder_params(cfg))
        model.load_state_dict({k_: torch.from_numpy(v) for k_, v in synth.make_state_dict(cfg, seed=42).items()})
        model = model.to(dev).train()
        from img2latex_amd.training import TrainStep
        ts = TrainStep(model, seed=1, overlap_all_reduce=not args.dp_single_allreduce, side_wgrad=not args.no_side_wgrad)
        images = torch.from_numpy(synth.make_images(Bt, cfg, seed=1234 + rank)).to(dev)
        forms = torch.from_numpy(synth.make_formulas(Bt, T, cfg["vocab_size"], seed=777 + rank)).to(torch.int32).to(dev)

        def one_step():
            ts.step(images, forms)
        unit = lambda: float(Bt * (T - 1))

        def roof(ms):       # 3 x forward flops = (433.85 + 149 x 1.835) MFLOP x 3 per sample
            fl = 3.0 * (433.85e6 + (T - 1) * 1.835e6) * Bt
            return dict(bound="mfma", kernel="whole step (conv / linear / LSTM forward + backward, CE, clip + Adam)",
                        achieved=round(fl / ms / 1e9, 2), peak=PEAK_FP32_TFLOPS, unit="TFLOP/s",
                        frac=round(fl / ms / 1e9 / PEAK_FP32_TFLOPS, 4), traffic=None,
                        note="algorithmic fp32 flops of the reference's step over the whole step time; GEMM-shaped work runs "
                             "as 3 x bf16 split products on the bf16 matrix cores, recurrences on the vector ALUs")

        def cpu_base():
            import img2latex_oracle as O
            cfg0 = synth.model_config(dropout=0.0)
            sd = O.to_torch_sd(synth.make_state_dict(cfg0, seed=42))
            x = torch.from_numpy(synth.make_images(Bt, cfg0, seed=1234))
            f = torch.from_numpy(synth.make_formulas(Bt, T, cfg0["vocab_size"], seed=777))
            st, ts_ = {}, []
            for _ in range(2):
                t0 = time.perf_counter()
                O.train_step(sd, cfg0, x, f, st)
                ts_.append(time.perf_counter() - t0)
            return {"value": round(Bt * (T - 1) / min(ts_), 1), "unit": "tokens/s", "cores": torch.get_num_threads(),
                    "kind": "port", "sample": f"the full per-GPU workload (B={Bt}, T={T - 1}), autograd fwd+bwd+clip+Adam, "
                                              "dropout off, best of 2"}
        name, conf = "training target tokens/sec (fwd+bwd+CE+clip+Adam)", {"workload": "cnn_lstm training step (BASELINE configs[3])", "batch_per_gpu": Bt, "global_batch": Bt * world, "seq_len": T, "dropout": 0.1, "backward_streams": ("1 (--no-side-wgrad)" if args.no_side_wgrad else "3: data-gradient chain on the step's stream, weight gradients on two side streams owned by TrainStep (i2l_lanes), joined before clip + Adam"), "parallelism": f"dp{world}: one flat-buffer all-reduce" + (" (single call)" if args.dp_single_allreduce else " (issued in two pieces, the first beside the conv backward)")}

    n_settle = 0
    if args.settle_ms > 0:                      # untimed: the same step until the GPU's clocks have ramped up from idle
        t_s = time.perf_counter()
        # a data-parallel training step holds a collective: every rank must run the SAME number of steps (fixed count)
        fixed = int(args.settle_ms / 2.0) if (dist is not None and args.mode == "train") else None
        while (n_settle < fixed) if fixed is not None else ((time.perf_counter() - t_s) * 1e3 < args.settle_ms):
            one_step()
            n_settle += 1
            if n_settle % 4 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        conf["settle"] = (f"{n_settle} untimed steps ({round((time.perf_counter() - t_s) * 1e3, 1)} ms) before the {args.warmup} warm-up "
                          "steps: clock ramp after idle (profiles/r03/ramp.txt); --settle-ms 0 turns it off")
    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    elapsed = time.perf_counter() - t0
    units = unit()
    serial_value_note = None
    if args.mode == "beam" and not args.serial:
        # batches are independent: the encoder of batch i + 1 runs on a second stream beside the beam search of batch i
        # (every batch still runs its own encoder and its own search; results checked below against the reference fixture)
        enc_stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)

        def enqueue_encoder():
            enc_stream.wait_stream(main)
            with torch.no_grad(), torch.cuda.stream(enc_stream):
                e = model.encoder(images)
                ev = torch.cuda.Event()
                ev.record(enc_stream)
            return e, ev

        def run_pipe(n_):
            nxt = enqueue_encoder()
            for _ in range(n_):
                e, ev = nxt
                main.wait_event(ev)
                e.record_stream(main)
                nxt = enqueue_encoder()
                with torch.no_grad():
                    out[0] = model.beam_search_batch(e, synth.START, synth.END, T, k)
            main.wait_event(nxt[1])                  # the look-ahead encoder of the batch after the last one
        run_pipe(args.warmup)
        fence()
        t0 = time.perf_counter()
        run_pipe(args.steps)
        fence()
        piped = time.perf_counter() - t0
        units = unit()
        conf["value_serial"] = round(units * args.steps / elapsed, 1)
        conf["value_pipelined"] = round(units * args.steps / piped, 1)
        conf["batch_pipeline"] = ("2 streams: encoder(i+1) beside the beam search of batch i (one extra look-ahead encoder per "
                                  "timed region is inside the time)" if piped < elapsed else "serial")
        elapsed = min(elapsed, piped)
    if args.mode == "resnet" and not args.serial:
        # the same co-resident two-stream schedule as the headline: decode(i) on the grouped co-resident kernel beside the trunk of
        # batch i + 1 (the ring GEMMs with a 2-stage ring fit beside it, the 4-stage ones wait); the serial pass above keeps
        # providing the encoder's time for the roofline and is a floor for `value`
        n_enc = args.pipe_encoders or 2
        pipe = GreedyPipeline(model, synth.START, synth.END, T, rows_per_workgroup=0, decode_flags=args.group_flag,
                              decode_priority=args.decode_priority, encoder_streams=n_enc,
                              hold_encoder=False if args.pipe_no_hold else None, clear_early=not args.pipe_late_clear)

        piped_ids = [None]

        def run_pipe(n):
            for _ in range(n):
                if pipe.pending() >= pipe.depth:
                    piped_ids[0] = pipe.collect()
                pipe.submit(images)
            while pipe.pending():
                piped_ids[0] = pipe.collect()
        run_pipe(args.warmup)
        fence()
        t0 = time.perf_counter()
        run_pipe(args.steps)
        fence()
        piped = time.perf_counter() - t0
        serial_value_note = round(units * args.steps / elapsed, 1)
        conf["value_serial"] = serial_value_note
        conf["value_pipelined"] = round(units * args.steps / piped, 1)
        conf["batch_pipeline"] = (f"{n_enc + 1} streams, co-resident: decode(i) on {args.group_kernel} beside the ResNet trunk(s) of the "
                                  f"next {n_enc} batch(es)"
                                  if piped < elapsed else "serial")
        elapsed = min(elapsed, piped)
        # outside the timed region: the pipelined batches decode to the ids of the serial pass (same images, same kernels
        # for the trunk; the 8-member decode against the 4-member one -- rows may differ only at fp32 near-ties)
        # ... run once more here with the trunk flags the pipeline uses (128-column ring tiles when two trunks are in flight:
        # another fp32 summation order, i.e. other bf16 roundings inside the trunk than the serial pass above)
        with torch.no_grad():
            saved_flags = model.encoder.kernel_flags
            model.encoder.kernel_flags = saved_flags | pipe.encoder_flags
            ref_ids, _ = model.greedy_ids(model.encoder(images), synth.START, synth.END, T)
            model.encoder.kernel_flags = saved_flags
        differ = int((piped_ids[0] != ref_ids.cpu()).any(dim=1).sum())
        conf["ids_check_pipelined"] = f"{Bn - differ} of {Bn} rows equal a serial pass with the pipeline's trunk kernels"
        if differ > Bn // 50:
            raise SystemExit(f"bench: pipelined resnet pass disagrees with the serial pass on {differ} rows")
    if dist is not None:
        tmax = torch.tensor([elapsed, 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0].item())
        tot = torch.tensor([units], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        units = float(tot.item())
    if args.mode == "train" and dist is not None:
        # the gradient all-reduce alone (the flat buffer [grads, loss sum, count]: 46.5 MB), outside the
        # timed region, HIP events on the stream the collective is enqueued from; bus bandwidth = 2 (N-1)/N x bytes / t
        # (what each xGMI link pair carries in a ring / the direct reduce-scatter + all-gather schedule)
        buf = torch.zeros_like(ts.flat_grads)
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 20
        ev[0].record()
        for _ in range(reps):
            dist.all_reduce(buf)
        ev[1].record()
        torch.cuda.synchronize()
        ar = torch.tensor([ev[0].elapsed_time(ev[1]) / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        ar_ms, nbytes = float(ar.item()), buf.numel() * 4
        conf.update(all_reduce_ms=round(ar_ms, 4), all_reduce_payload_mb=round(nbytes / 1e6, 2),
                    all_reduce_algbw_gbs=round(nbytes / ar_ms / 1e6, 1),
                    all_reduce_busbw_gbs=round(2.0 * (world - 1) / world * nbytes / ar_ms / 1e6, 1),
                    all_reduce_backend=dist.get_backend(),
                    all_reduce_note="alone on the chip; inside the step it is issued in two pieces and the first "
                                    "(99 % of the bytes) overlaps the conv backward (dp.OverlappedAllReduce)")
    if args.mode == "resnet":
        ms = float(np.median(conf["encoder_ms"][-args.steps:]))
        conf["encoder_ms"] = round(ms, 4)
        conf["encoder_images_per_s"] = round(args.batch / ms * 1e3, 1)
        conf["encoder_tflops"] = round(conf["encoder_gflop"] / ms, 2)
        conf["encoder_frac_of_bf16_dense_peak_2500TF"] = round(conf["encoder_gflop"] / ms / 2500.0, 4)
    if rank == 0:
        line = {"metric": name, "value": round(units * args.steps / elapsed, 1), "unit": conf.pop("unit", "tokens/s"),
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.mode == "resnet" else ("u8" if args.mode == "preprocess" else ("int32" if args.mode == "metrics" else "f32")),
                "data": "synthetic", "config": conf,
                "roofline": roof(elapsed / args.steps * 1e3)}
        if args.mode == "beam":
            line["ids_check"] = after_check()
        if world == 1 and not args.no_cpu_baseline and cpu_base is not None:
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            torch.set_num_threads(host_cores())
            line["cpu_baseline"] = cpu_base()
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def host_cores():
    """CPU share of this process: affinity mask, capped by the cgroup quota and by the GPU
    box's documented share (16 cores per GPU) -- os.cpu_count() reports the whole host and
    oversubscribing OpenMP by 10x stalls the oracle."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(cfg, sd_kw, B, T):
    """The CPU oracle (id-exact restatement of the reference's PyTorch CPU path, pinned by
    tests/golden) timed on this host's cores on the SAME workload: full batch, full 150 steps,
    median of 3 runs (about 10-20 s of CPU work)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import img2latex_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu baseline on {cores} threads ...", file=sys.stderr, flush=True)
    sd = O.to_torch_sd(synth.make_state_dict(cfg, **sd_kw))
    x = torch.from_numpy(synth.make_images(B, cfg, seed=1234))
    times, enc_t, steps = [], [], T
    with torch.no_grad():
        for _ in range(3):
            t0 = time.perf_counter()
            enc = O.cnn_encoder(sd, cfg, x)
            t1 = time.perf_counter()
            ids = O.greedy_search(sd, cfg, enc, synth.START, synth.END, T)
            t2 = time.perf_counter()
            steps = len(ids[0]) - 1
            times.append(t2 - t0)
            enc_t.append(t1 - t0)
            print(f"[bench] cpu run {len(times)}: {t2 - t0:.2f} s", file=sys.stderr, flush=True)
            if sum(times) > 30.0:
                break
    med = float(np.median(times))
    return {"value": round(B * steps / med, 1), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"full workload: B={B}, {steps} steps, median of {len(times)} (encoder {np.median(enc_t) * 1e3:.0f} ms of "
                      f"{med * 1e3:.0f} ms), torch {torch.__version__} CPU fp32, {cores} threads"}


if __name__ == "__main__":
    main()
