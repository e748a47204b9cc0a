"""Where the host time of one replayed step goes (batch 4: the GPU is always ahead, so wall time between launches is host work):
per phase, and inside the C replay (segments, entries, time in nkb_plan_run)."""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench
from nkb_classification import hip
args = bench.parse()
dev = torch.device("cuda:0")
model, opt, crit = bench.build(args, dev)
img = torch.randn(4, 3, 224, 224).to(dev); tgt = torch.randint(0, args.classes, (4,)).to(dev)
model.train()
amp = args.dtype in ("bf16", "fp8")
model.fp8_linear = args.dtype == "fp8"
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
def step(measure):
    t = time.perf_counter(); opt.zero_grad(); measure and tick("zero_grad", t)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        t = time.perf_counter(); out = model(img); measure and tick("forward", t)
        t = time.perf_counter(); loss = crit(out, tgt); measure and tick("loss", t)
    t = time.perf_counter(); loss.backward(); measure and tick("backward", t)
    t = time.perf_counter(); opt.step(); measure and tick("optimizer", t)
for _ in range(8): step(False)
torch.cuda.synchronize()
N = 30
for _ in range(N):
    step(True); torch.cuda.synchronize()
print("host ms per step by phase:", {k: round(1e3 * v / N, 3) for k, v in T.items()}, "total", round(1e3 * sum(T.values()) / N, 3))
eng = model._active
for key, ent in eng.plans.items():
    plan = ent[0]
    kinds = [s[0] for s in plan.segments]
    nc = sum(s[2] for s in plan.segments if s[0] == 0)
    print("plan", str(key)[:60], "segments", len(kinds), "C entries", nc, "py ops", kinds.count(2), "legacy calls", kinds.count(1))
