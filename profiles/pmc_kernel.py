"""Per-kernel averages of the PMC counters in a rocprofv3 counter_collection csv (arg 1), kernels matching arg 2."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if pat in k:
        key = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60] + " grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
