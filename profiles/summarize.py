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
    "decode_group8_kernel": ("decode8", False),          # the pipelined region's decode (8 members x 8 rows)
    "decode_group16_kernel": ("decode16", False),        # r04: 16 members x 16 rows on the matrix cores
    "conv3x3_smallk_bf16x3_kernel": ("conv0", False), "conv3x3_bf16x3_kernel<2,": ("conv1", False),
    "conv3x3_bf16x3_kernel<1,": ("conv2", False), "conv3x3_bf16x3_kernel<4,": ("conv2", False),
    "gemm_bf16x3_kernel<true, true,": ("fc", True), "splitk_reduce_kernel": ("fc_reduce", False),
}

def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}

out = {"_note": __doc__.split("\n\n")[1].replace("\n", " "),
       "_collection": f"profiles/{rnd}/pmc_fetch_size_{tag}.csv + pmc_write_size_{tag}.csv (rocprofv3 --pmc passes of `python bench.py`, {rnd} {tag})",
       "raw_kb": {}, "_kernels": {}}
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
# matrix-pipe utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the
# 8 XCDs (MI355X_MICROARCH.md): util = busy / (GUI_ACTIVE / 8 * 1024); clock = GUI_ACTIVE / 8 / kernel time
def mfma_table(sub, dst_name):
    path = f"{src}/{sub}/p_counter_collection.csv"
    if not os.path.exists(path):
        return
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            acc[row["Kernel_Name"]]["ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    rows = []
    for k, d in acc.items():
        busy, gui = d.get("SQ_VALU_MFMA_BUSY_CYCLES"), d.get("GRBM_GUI_ACTIVE")
        if not busy or not gui or sum(busy) == 0:
            continue
        b, g = sum(busy) / len(busy), sum(gui) / len(gui)
        ns = sum(d["ns"]) / len(d["ns"])
        rows.append([k[:110], len(busy), ns / 1e3, b, g, b / (g / 8 * 1024), g / 8 / ns])
    if not rows:
        return
    # GRBM_GUI_ACTIVE is not kernel-exclusive: it also counts the dispatch window around the kernel, so for kernels of
    # tens of microseconds "GUI_ACTIVE / 8 / kernel time" reads 2.9-3.5 GHz on a part that runs <= 2.4 GHz and
    # busy / GUI understates the utilisation by an unknown factor.  The reference clock is
    # therefore taken from the LONGEST kernel of the collection (>= 200 us: the window error is < 2 %), and the
    # time-based utilisation busy / (1024 SIMDs x kernel time x that clock) is reported beside the GUI-based one;
    # rows whose own GUI clock is implausible (> 2.45 GHz) are flagged and only their time-based figure is meaningful
    # (it still assumes the long kernel's clock: a short non-MFMA kernel may clock higher, which only lowers its figure).
    longest = max(rows, key=lambda r: r[2])
    clk_ref = min(longest[6], 2.4) if longest[2] >= 200.0 else 2.1
    with open(f"{dst}/{dst_name}", "w") as f:
        f.write(f"# clock reference {clk_ref:.3f} GHz from the longest kernel ({longest[0][:60]}, {longest[2]:.0f} us)\n")
        f.write("kernel,launches,avg_us_under_pmc,mfma_busy_cycles,gui_active,mfma_util_gui_window,clock_ghz_gui_window,"
                "mfma_util_time_based,gui_window_exclusive\n")
        for r in sorted(rows, key=lambda r: -r[2]):
            util_t = r[3] / (1024.0 * r[2] * 1e3 * clk_ref)
            if sub == "mfma":                       # the headline's kernels: matrix-pipe busy fraction per bench stage (bench.py reads it)
                for frag, (stage, _) in STAGE.items():
                    if frag in r[0]:
                        out.setdefault("_mfma_busy", {})[stage] = round(util_t, 4)
            f.write(f"\"{r[0]}\",{r[1]},{r[2]:.1f},{r[3]:.0f},{r[4]:.0f},{r[5]:.4f},{r[6]:.3f},{util_t:.4f},"
                    f"{'yes' if r[6] <= 2.45 else 'NO'}\n")
    print(dst_name, f"(clock reference {clk_ref:.2f} GHz)")
    for r in sorted(rows, key=lambda r: -r[2])[:8]:
        print(f"  {r[0][:70]:70s} {r[2]:8.1f} us  util(gui) {r[5]:.3f}  util(time) {r[3] / (1024.0 * r[2] * 1e3 * clk_ref):.3f}  gui clock {r[6]:.2f} GHz")

mfma_table("mfma", f"pmc_mfma_busy_{tag}.csv")
mfma_table("mfma_train", f"pmc_mfma_busy_train_{tag}.csv")
mfma_table("mfma_resnet", f"pmc_mfma_busy_resnet_{tag}.csv")
# secondary configs: bench lines (one JSON line per mode) and per-kernel stats
lines = [open(f"{src}/bench.json").read().strip()]
for m in ("beam", "train", "resnet", "preprocess", "metrics"):
    if os.path.exists(f"{src}/bench_{m}.json"):
        lines.append(open(f"{src}/bench_{m}.json").read().strip())
    st = f"{src}/stats_{m}/p_kernel_stats.csv"
    if os.path.exists(st):
        shutil.copy(st, f"{dst}/{m}_{tag}_kernel_stats.csv")
with open(f"{dst}/bench_all_modes_{tag}.jsonl", "w") as f:
    f.write("\n".join(l for l in lines if l) + "\n")
if os.path.exists("gpurun_out/parity_errors.json"):
    shutil.copy("gpurun_out/parity_errors.json", f"{dst}/parity_errors_{tag}.json")
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_") and k != "raw_kb"}, indent=1))
