"""Every launch of the LAST (serialized) train step of a rocprofv3 kernel-trace CSV, in start order: start offset (us), duration
(us), grid size, short kernel name.  Usage: python scripts/step_dump.py <kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?")))
rows.sort()
opt = [i for i, r in enumerate(rows) if "optim_step_kernel" in r[2]]
ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] != i + 1]
sel = rows[ends[-2] + 1:ends[-1] + 1]
t0 = sel[0][0]
for a, b, name, grid, wg in sel:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{(a - t0) / 1e3:10.1f} {(b - a) / 1e3:8.1f} {grid:>9s} {wg:>5s}  {short[:90]}")
