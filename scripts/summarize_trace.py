"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid) -> launches, avg/total microseconds."""
import csv, sys, collections
rows = collections.defaultdict(lambda: [0, 0.0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0][:60]
        key = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))
        d = rows[key]
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tot = sum(v[1] for v in rows.values())
print(f"total kernel time {tot/1e3/steps:.3f} ms/step over {steps} steps")
for k, v in sorted(rows.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{k[0]:60s} wg={k[1]:6d} y={k[2]:4d} n/step={v[0]/steps:6.1f} avg={v[1]/v[0]:9.1f}us total/step={v[1]/steps/1e3:7.3f}ms")
