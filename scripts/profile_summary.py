"""Per-kernel summary of the LAST `keep` train steps in a rocprofv3 kernel-trace CSV (bench.py runs its HIP-event
profiled steps last, with the weight-gradient stream folded onto the main stream, so these are the launches the
`roofline` object of the bench line is computed from).  Step boundaries = launches of optim_step_kernel.
Usage: python scripts/profile_summary.py <kernel_trace.csv> <keep> <out.csv>"""
import csv, sys, collections

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
keep = int(sys.argv[2])
opt_idx = [i for i, r in enumerate(rows) if r[2].startswith("optim_step_kernel") or "optim_step_kernel" in r[2]]
# one optimizer launch per parameter group per step: group consecutive optimizer launches
ends = [i for k, i in enumerate(opt_idx) if k + 1 == len(opt_idx) or opt_idx[k + 1] != i + 1]
first = ends[-keep - 1] + 1 if len(ends) > keep else 0
sel = rows[first:ends[-1] + 1]
agg = collections.OrderedDict()
for s, e, name in sel:
    short = name.replace("(anonymous namespace)::", "").split("(")[0]
    a = agg.setdefault(short, [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(v[1] for v in agg.values())
with open(sys.argv[3], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls_per_step", "avg_us", "total_ms_per_step", "percent"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, round(v[0] / keep, 2), round(v[1] / v[0] / 1e3, 2), round(v[1] / keep / 1e6, 4), round(100.0 * v[1] / tot, 2)])
print(f"{len(sel)} launches in the last {keep} steps, {tot / keep / 1e6:.3f} ms of kernel time per step -> {sys.argv[3]}")
