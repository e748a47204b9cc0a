"""Per-shape floors of one train step (VERDICT r2 'top_next': kernels within 1.5x of their per-shape floors): every profiled launch
of the step (HIP events, weight-gradient stream folded onto the main stream, as bench.py's roofline leg) grouped by (kernel family,
algorithmic FLOPs, algorithmic bytes), with its floor = max(FLOPs / MFMA rate, bytes / HBM rate) at the PRACTICAL rates of this chip
(1.2 PFLOP/s bf16 for tiles of this size — what the eight-phase core reaches on long K; 5.5 TB/s — what the streaming kernels reach)
and at the PEAK rates (2.5 PFLOP/s, 8 TB/s).  Sorted by the time above the practical floor.
Caveat: weight-gradient launches are sized to run BESIDE the main stream (128-256 workgroups, §4 of DESIGN.md), so their serialized
times here are longer than what they cost the overlapped step; the data-path rows (conv_igemm_*) are the ones to read as "x floor".
Usage: python scripts/floor_table.py [bench.py arguments]"""
import collections, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench
from nkb_classification import hip
args = bench.parse()
dev = torch.device("cuda:0")
model, opt, crit = bench.build(args, dev)
model.fp8_linear = args.dtype == "fp8"
hs = bench.head_sizes(args)
g = torch.Generator().manual_seed(1)
img = torch.randn(args.batch, 3, 224, 224, generator=g).to(dev); tgt = bench.make_targets(hs, args.classes, args.batch, g, dev)
model.train()
def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype != "f32"):
        loss = crit(model(img), tgt)
    (loss["loss"] if hs else loss).backward(); opt.step()
for _ in range(4): step()
torch.cuda.synchronize()
engines = list(getattr(model, "_engines", {}).values())
for e in engines: e.overlap_wgrad = False
for _ in range(2): step()
torch.cuda.synchronize()
hip.prof_enable(True)
N = 5
for _ in range(N): step()
torch.cuda.synchronize(); hip.prof_enable(False)
recs = hip.prof_collect_raw(with_bytes=True)
agg = collections.defaultdict(lambda: [0, 0.0])
for name, ms, work, byt in recs:
    d = agg[(name, work, byt)]; d[0] += 1; d[1] += ms
MF, HB, MFP, HBP = 1.2e15, 5.5e12, (5.0e15 if args.dtype == "fp8" else 2.5e15), 8.0e12
rows = []
for (name, work, byt), (cnt, ms) in agg.items():
    avg = 1e3 * ms / cnt
    fl = 1e6 * max(work / MF, byt / HB); flp = 1e6 * max(work / MFP, byt / HBP)
    rows.append((cnt / N * max(avg - fl, 0.0) if fl > 0 else 0.0, name, work, byt, cnt / N, avg, fl, flp))
rows.sort(reverse=True)
tot = sum(r[4] * r[5] for r in rows); above = sum(r[0] for r in rows if r[6] >= 2.0)
print(f"{args.model} bs {args.batch} {args.dtype}: {tot / 1e3:.2f} ms of kernel time per step (serialized), {above / 1e3:.2f} ms above the practical floors of the launches that declare work")
print(f"{'kernel':18s} {'GFLOP':>8s} {'MB':>7s} {'n/step':>6s} {'avg us':>8s} {'floor us':>9s} {'x floor':>7s} {'peak-floor us':>13s} {'above, us/step':>14s}")
for ex, name, work, byt, n, avg, fl, flp in rows:
    if fl < 2.0: continue              # (launches that declare a token amount of work: small algebra, finalizes)
    print(f"{name:18s} {work / 1e9:8.2f} {byt / 1e6:7.1f} {n:6.1f} {avg:8.1f} {fl:9.1f} {avg / fl:7.2f} {flp:13.1f} {ex:14.1f}")
