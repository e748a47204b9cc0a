"""Per-launch device times of ONE serialized train step (library HIP-event profiler), grouped by kernel tag and sorted."""
import os, sys, torch, argparse, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench
from nkb_classification import hip
args = argparse.Namespace(model=sys.argv[1] if len(sys.argv) > 1 else "resnet50", classes=1000, batch=int(os.environ.get("BS", 256)))
dev = torch.device("cuda", 0)
model, opt, crit = bench.build(args, dev)
g = torch.Generator().manual_seed(1)
img = torch.randn(args.batch, 3, 224, 224, generator=g).to(dev); tgt = torch.randint(0, 1000, (args.batch,), generator=g).to(dev)
model.train()
model.fp8_linear = os.environ.get("NKB_FP8") == "1"
def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = crit(model(img), tgt)
    loss.backward(); opt.step()
for _ in range(3): step()
for e in model._engines.values(): e.overlap_wgrad = False
for _ in range(3): step()
torch.cuda.synchronize()
hip.prof_enable(True); step(); torch.cuda.synchronize(); hip.prof_enable(False)
raw = hip.prof_collect_raw()
tot = collections.defaultdict(float)
for k, ms, w in raw: tot[k] += ms
print("total ms", round(sum(tot.values()), 2), {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
for i, (k, ms, w) in enumerate(raw):
    if ms > float(os.environ.get("MIN_MS", 0.06)): print(f"{i:4d} {k:18s} {ms*1e3:8.1f} us  {w/1e9:8.1f} GF  {w/ms/1e9 if ms else 0:7.0f} TF/s")
