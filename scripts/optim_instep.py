"""The optimizer step of a built unicom ViT-L/14 model, timed alone (after one train step has created its state): is the in-step
3.1 ms (2.9 TB/s) a property of the arena's memory or of what runs around it?"""
import os, sys, types, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench
args = types.SimpleNamespace(model="unicom ViT-L/14", classes=1000)
dev = torch.device("cuda:0")
model, opt, crit = bench.build(args, dev)
model.train()
x = torch.randn(8, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (8,), device=dev)
for _ in range(2):
    opt.zero_grad(set_to_none=True); loss = crit(model(x), y); loss.backward(); opt.step()
torch.cuda.synchronize()
a = model.arena if hasattr(model, "arena") else None
for name in ("flat_param", "flat_grad"):
    t = getattr(a, name, None)
    if t is not None: print(name, hex(t.data_ptr()), t.numel())
m, v = a.moments(); print("m", hex(m.data_ptr()), "v", hex(v.data_ptr()), "shadow", hex(a.shadow.data_ptr()) if a.shadow is not None else None)
for trial in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): opt.step()
    e1.record(); torch.cuda.synchronize()
    print("opt.step alone: %.3f ms" % (e0.elapsed_time(e1) / 5))
