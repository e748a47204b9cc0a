"""Where the wall time of an OVERLAPPED train step goes (rocprofv3 kernel-trace CSV of bench.py).

bench.py's timed steps run the weight gradients on a side stream; its last 5 steps are serialized for the roofline events.  This
script takes the overlapped steps just before those and reports, per step: wall span, per-queue busy time, time during which NO
kernel was running (launch gaps / dependency bubbles), time with >= 2 kernels running, and per kernel family the average duration
in the overlapped steps next to the serialized ones (the stretch is what co-running costs that kernel).
Usage: python scripts/overlap_report.py <kernel_trace.csv> [serialized_steps=5] [overlapped_steps=5]"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
ser = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ovl = int(sys.argv[3]) if len(sys.argv) > 3 else 5
opt = [i for i, r in enumerate(rows) if "optim_step_kernel" in r[2]]
ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] != i + 1]


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def steps(lo, hi):
    return [rows[ends[k] + 1:ends[k + 1] + 1] for k in range(lo, hi)]


n = len(ends)
ser_steps = steps(n - 1 - ser, n - 1)
ovl_steps = steps(n - 1 - ser - ovl, n - 1 - ser)


def avg_by_kernel(stp):
    d = collections.defaultdict(lambda: [0, 0])
    for s in stp:
        for a, b, name, _ in s:
            e = d[short(name)]
            e[0] += 1
            e[1] += b - a
    return d


def timeline(s):
    ev = []
    for a, b, _, _ in s:
        ev.append((a, 1))
        ev.append((b, -1))
    ev.sort()
    t0, t1 = s[0][0], max(x[1] for x in s)
    idle = multi = 0
    depth, last = 0, t0
    for t, d in ev:
        if depth == 0:
            idle += t - last
        elif depth >= 2:
            multi += t - last
        depth += d
        last = t
    return (t1 - t0) / 1e6, idle / 1e6, multi / 1e6


print("overlapped steps: wall ms, idle ms (no kernel running), ms with >= 2 kernels running, per-queue busy ms")
for s in ovl_steps:
    wall, idle, multi = timeline(s)
    q = collections.defaultdict(int)
    for a, b, _, qid in s:
        q[qid] += b - a
    print(f"  wall {wall:7.3f}  idle {idle:6.3f}  multi {multi:6.3f}  queues " + "  ".join(f"{k}: {v / 1e6:.3f}" for k, v in sorted(q.items())))
print("serialized steps:")
for s in ser_steps:
    wall, idle, multi = timeline(s)
    print(f"  wall {wall:7.3f}  idle {idle:6.3f}  kernel sum {sum(b - a for a, b, _, _ in s) / 1e6:7.3f}")
A, B = avg_by_kernel(ovl_steps), avg_by_kernel(ser_steps)
print(f"{'kernel':70s} {'n/step':>6s} {'ser us':>8s} {'ovl us':>8s} {'stretch':>7s} {'ovl ms/step':>11s}")
for k, (cnt, tot) in sorted(A.items(), key=lambda kv: -kv[1][1])[:40]:
    sc, st = B.get(k, (0, 0))
    su = st / sc / 1e3 if sc else 0.0
    ou = tot / cnt / 1e3
    print(f"{k[:70]:70s} {cnt / len(ovl_steps):6.1f} {su:8.1f} {ou:8.1f} {ou / su if su else 0:7.2f} {tot / len(ovl_steps) / 1e6:11.3f}")
