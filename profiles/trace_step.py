import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
tot = 0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    tot += d
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print(f"{name:44s} grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) {d:8.1f} us")
print("total", tot)
