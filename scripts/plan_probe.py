"""Host-time breakdown of the train step with and without recorded launch plans (bs 256 ResNet-50 / ViT-B/16, bf16)."""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import argparse
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench
args = argparse.Namespace(model=sys.argv[1] if len(sys.argv) > 1 else "resnet50", classes=1000, batch=256)
dev = torch.device("cuda", 0)
model, opt, crit = bench.build(args, dev)
g = torch.Generator().manual_seed(1)
BS = int(sys.argv[2]) if len(sys.argv) > 2 else 256
img = torch.randn(BS, 3, 224, 224, generator=g).to(dev); tgt = torch.randint(0, 1000, (BS,), generator=g).to(dev)
model.train()
def step(tm):
    t0 = time.perf_counter(); opt.zero_grad(); t1 = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        preds = model(img); t2 = time.perf_counter()
        loss = crit(preds, tgt)
    t3 = time.perf_counter(); loss.backward(); t4 = time.perf_counter(); opt.step(); t5 = time.perf_counter()
    for k, v in zip(("zero", "fwd", "loss", "bwd", "opt"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): tm[k] = tm.get(k, 0) + v
for i in range(8):
    step({}); torch.cuda.synchronize()
    eng = next(iter(model._engines.values()))
    print(i, "gen", eng.ws.generation, "plans", [k[0] for k in eng.plans], flush=True)
tm = {}
for i in range(20): step(tm)
torch.cuda.synchronize()
print({k: round(v / 20 * 1e3, 2) for k, v in tm.items()}, "ms per step (host)")
