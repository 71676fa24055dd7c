import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# counter_collection.csv: Kernel_Name, Counter_Name, Counter_Value, Grid_Size..., Dispatch_Id
by = collections.OrderedDict()
for r in rows:
    key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", ""))
    by.setdefault(key, {})[r["Counter_Name"]] = by.get(key, {}).get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k, v in by.items():
    if "ring" in k[1] or "gemm_bf16" in k[1]:
        print(k[0], k[1][-28:], k[2], " ".join(f"{n}={x:.0f}" for n, x in v.items()))
