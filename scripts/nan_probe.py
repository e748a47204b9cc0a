"""Debugging aid: trains the ResNet-50 bench configuration and stops at the first step whose logits, loss or any parameter gradient is
non-finite, naming the tensors (round 3: pointed at layer1.*.conv2.weight = wgrad3x3_kernel).  Usage: python scripts/nan_probe.py <steps> <runs>
(NKB_WGRAD_STREAM=0 NKB_PLAN=0 make the step single-stream and eager)."""
import argparse, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench
def run(steps, tag):
    args = argparse.Namespace(model="resnet50", classes=1000, batch=256, dtype="bf16", heads="")
    dev = torch.device("cuda:0")
    model, opt, crit = bench.build(args, dev)
    g = torch.Generator().manual_seed(7)
    img = torch.randn(256, 3, 224, 224, generator=g).to(dev); tgt = torch.randint(0, 1000, (256,), generator=g).to(dev)
    model.train()
    for i in range(steps):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(img)
            loss = crit(out, tgt)
        loss.backward()
        bad = [(n, int((~torch.isfinite(p.grad)).sum())) for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        if bad or not torch.isfinite(out).all() or not torch.isfinite(loss):
            print(f"{tag}: step {i}: logits finite {bool(torch.isfinite(out).all())} loss {loss.item()} non-finite grads in {len(bad)} tensors: {bad[:12]}", flush=True)
            bufs = [(n, int((~torch.isfinite(b)).sum())) for n, b in model.named_buffers() if b.is_floating_point() and not torch.isfinite(b).all()]
            print(f"   non-finite buffers: {bufs[:8]}", flush=True)
            return True
        opt.step()
    return False
n, k = int(sys.argv[1]), int(sys.argv[2])
hits = sum(run(n, f"run {j}") for j in range(k))
print(f"{k} runs x {n} steps: {hits} runs hit a non-finite value", flush=True)
