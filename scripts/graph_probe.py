"""Feasibility probe: capture one whole train step (forward, loss, backward, fused optimizer) in a HIP graph through
torch.cuda.CUDAGraph and time replays against eager launches.  Optimizer scalars are frozen at capture, so this is a
timing experiment only."""
import os, sys, time, torch
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
    return loss
for _ in range(6): step()
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager  %.3f ms/step" % timeit(step))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, stream=s):
        loss = step()
    torch.cuda.synchronize()
    print("graph  %.3f ms/step" % timeit(g.replay))
    print("eager  %.3f ms/step" % timeit(step))
except Exception as e:
    import traceback; traceback.print_exc()
    print("capture failed:", type(e).__name__, str(e)[:300])
