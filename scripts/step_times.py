"""Per-step host-enqueue and synchronised wall time of the first N train steps (warm-up behaviour)."""
import sys, os, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench
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
SYNC = os.environ.get("STEP_SYNC", "1") != "0"
T0 = time.perf_counter()
for i in range(16):
    if SYNC: torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    if SYNC: torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {i:2d}: enqueue {1e3*(t1-t0):7.1f} ms  total {1e3*(t2-t0):7.1f} ms  mem {torch.cuda.memory_allocated()/2**30:.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:.2f}")
torch.cuda.synchronize(); print("all 16 steps: %.1f ms" % (1e3 * (time.perf_counter() - T0)))
