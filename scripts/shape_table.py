"""Per-shape MFMA efficiency of the conv kernels on one ResNet-50 bf16 train step (HIP-event profiler)."""
import sys, collections, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench
from nkb_classification import hip
args = bench.parse()
dev = torch.device("cuda:0")
model, opt, crit = bench.build(args, dev)
img = torch.randn(args.batch, 3, 224, 224).to(dev); tgt = torch.randint(0, args.classes, (args.batch,)).to(dev)
model.train()
def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.dtype == "bf16"):
        loss = crit(model(img), tgt)
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
hip.prof_enable(True)
N = 3
for _ in range(N): step()
torch.cuda.synchronize(); hip.prof_enable(False)
recs = hip.prof_collect_raw()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, ms, work in recs:
    d = agg[(name, work)]; d[0] += 1; d[1] += ms
tot = sum(v[1] for v in agg.values()) / N
print(f"total {tot:.2f} ms/step")
for (name, work), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if work > 0:
        print(f"{name:18s} gflop/launch={work/1e9:8.2f} n/step={cnt/N:5.1f} avg={1e3*ms/cnt:8.1f}us  {work*cnt/ms/1e9:7.1f} TF/s  total={ms/N:6.3f} ms/step")
