"""Where the GPU idles inside an OVERLAPPED train step of a rocprofv3 kernel-trace CSV: every interval with no kernel running on any
queue, longest first, with the kernels before and after it; and the total of the short gaps (< 5 us) by the kernel that follows.
Usage: python scripts/idle_gaps.py <kernel_trace.csv> [serialized_steps_at_end=5]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48], r.get("Queue_Id", "?")))
rows.sort()
ser = int(sys.argv[2]) if len(sys.argv) > 2 else 5
opt = [i for i, r in enumerate(rows) if "optim_step_kernel" in r[2]]
ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] != i + 1]
# the step before the serialized tail
lo, hi = ends[-ser - 2] + 1, ends[-ser - 1] + 1
step = rows[lo:hi]
t0, t1 = step[0][0], max(r[1] for r in step)
gaps = []
cur_end, last = step[0][1], step[0][2]
for a, b, name, q in step[1:]:
    if a > cur_end: gaps.append((a - cur_end, last, name, (cur_end - t0) / 1e3))
    if b > cur_end: cur_end, last = b, name
tot = sum(g[0] for g in gaps)
print(f"step wall {(t1 - t0) / 1e3:.1f} us, {len(step)} launches, idle {tot / 1e3:.1f} us in {len(gaps)} gaps")
print("longest gaps (us, at us, after -> before):")
for g in sorted(gaps, reverse=True)[:15]: print(f"  {g[0] / 1e3:7.1f} at {g[3]:9.1f}  {g[1]} -> {g[2]}")
small = collections.Counter(); cnt = collections.Counter()
for g in gaps:
    if g[0] < 5000: small[g[2]] += g[0]; cnt[g[2]] += 1
print(f"gaps under 5 us: {sum(small.values()) / 1e3:.1f} us in {sum(cnt.values())} gaps; by following kernel:")
for k, v in small.most_common(12): print(f"  {v / 1e3:7.1f} us in {cnt[k]:4d}  {k}")
