#!/usr/bin/env python3
"""Turn gpurun_out/prof_<round>_<tag>/ (profiles/collect.sh) into the committed evidence:
   profiles/<round>/bench_<tag>.json, bench_<tag>_kernel_stats.csv, pmc_fetch_size_<tag>.csv,
   pmc_write_size_<tag>.csv (per-kernel averages) and profiles/traffic.json (HBM-side bytes per launch,
   read by bench.py for roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes, i.e. reports half the bytes of wide (16 B/lane) streams -> kernels whose loads are all 16-byte are
doubled; other widths are uncalibrated and reported raw (flagged).  WRITE_SIZE is exact."""
import csv, json, os, shutil, sys
from collections import defaultdict

rnd, tag = (sys.argv + ["r01", "v3"])[1:3]
src = f"gpurun_out/prof_{rnd}_{tag}"
dst = f"profiles/{rnd}"
os.makedirs(dst, exist_ok=True)
shutil.copy(f"{src}/bench.json", f"{dst}/bench_{tag}.json")
shutil.copy(f"{src}/stats/p_kernel_stats.csv", f"{dst}/bench_{tag}_kernel_stats.csv")

STAGE = {  # kernel-name fragment -> bench stage, all loads 16-byte wide?
    "decode_group_kernel": ("decode", False), "decode_kernel": ("decode", True),
    "conv3x3_smallk_bf16x3_kernel": ("conv0", False), "conv3x3_bf16x3_kernel<2>": ("conv1", False),
    "conv3x3_bf16x3_kernel<1>": ("conv2", False), "conv3x3_bf16x3_kernel<4>": ("conv2", False),
    "linear_bf16x3_kernel": ("fc", True), "splitk_reduce_kernel": ("fc_reduce", False),
}

def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

out = {"_note": __doc__.split("\n\n")[1].replace("\n", " "), "raw_kb": {}, "_kernels": {}}
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    avg = per_kernel(f"{src}/{sub}/p_counter_collection.csv", counter)
    with open(f"{dst}/pmc_{counter.lower()}_{tag}.csv", "w") as f:
        f.write("kernel,launches,avg_kb_per_launch\n")
        for k, (v, n) in sorted(avg.items(), key=lambda kv: -kv[1][0]):
            f.write(f"\"{k[:120]}\",{n},{v:.3f}\n")
    for k, (v, n) in avg.items():
        for frag, (stage, wide) in STAGE.items():
            if frag in k:
                out["raw_kb"].setdefault(stage, {})[counter] = v
                out["_kernels"][stage] = frag
for stage, d in out["raw_kb"].items():
    wide = next(w for f, (s, w) in STAGE.items() if s == stage and out["_kernels"][stage] == f)
    fetch = d.get("FETCH_SIZE", 0.0) * 1024 * (2 if wide else 1)
    out[stage] = int(fetch + d.get("WRITE_SIZE", 0.0) * 1024)
    out.setdefault("_fetch_doubled", {})[stage] = wide
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_") and k != "raw_kb"}, indent=1))
