"""Soak of a real bench-size train step (recorded plans replayed in C, weight gradients co-running on the second stream): K runs of N
steps from the same seed must produce bit-identical parameter checksums at EVERY step — the step has no atomics in its arithmetic,
so any difference is a race, a stale read or uninitialised memory.  Op-level tests run on fresh buffers and cannot see these.
(Round 3 found two this way: a one-launch form of the BatchNorm tile statistics that read the previous launch's partials from L2,
and wgrad3x3_kernel's 0 x NaN on uninitialised LDS — a non-finite layer1 gradient about once per 500 ResNet-50 steps.)
Usage: python scripts/soak_determinism.py [--steps 150] [--runs 2] [--model resnet50] [--batch 256] [--dtype bf16] [--heads ""] [--size 224]"""
import argparse, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench


def run(steps, model_name="resnet50", batch=256, dtype="bf16", heads="", size=224):
    """-> (final parameters, final float buffers, final loss, per-step parameter checksums)"""
    args = argparse.Namespace(model=model_name, classes=1000, batch=batch, dtype=dtype, heads=heads)
    dev = torch.device("cuda:0")
    model, opt, crit = bench.build(args, dev)
    model.fp8_linear = dtype == "fp8"
    hs = bench.head_sizes(args)
    g = torch.Generator().manual_seed(7)
    img = torch.randn(batch, 3, size, size, generator=g).to(dev)
    tgt = bench.make_targets(hs, 1000, batch, g, dev)
    model.train()
    torch.manual_seed(99)                      # stochastic-depth / dropout seed stream
    sums = []
    for _ in range(steps):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype != "f32"):
            loss = crit(model(img), tgt)
        loss = loss["loss"] if hs else loss
        loss.backward()
        opt.step()
        sums.append(model.arena.flat_param.double().sum())
    torch.cuda.synchronize()
    bufs = [b.detach().float().flatten() for b in model.buffers() if b.is_floating_point()]
    return model.arena.flat_param.clone(), (torch.cat(bufs).clone() if bufs else torch.zeros(1)), loss.item(), [s.item() for s in sums]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150); ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--model", default="resnet50"); ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="bf16"); ap.add_argument("--heads", default=""); ap.add_argument("--size", type=int, default=224)
    a = ap.parse_args()
    ref = run(a.steps, a.model, a.batch, a.dtype, a.heads, a.size)
    bad = 0
    for j in range(1, a.runs):
        r = run(a.steps, a.model, a.batch, a.dtype, a.heads, a.size)
        d = [i for i in range(a.steps) if not (ref[3][i] == r[3][i])]
        if d or not torch.equal(ref[0], r[0]) or not torch.equal(ref[1], r[1]):
            bad += 1
            print(f"  run {j}: first difference at step {d[0] if d else 'end'}: {ref[3][d[0]] if d else ''} vs {r[3][d[0]] if d else ''}", flush=True)
    import math
    print(f"{a.model} bs {a.batch} {a.dtype} {a.heads} {a.size}px: {a.runs} runs x {a.steps} steps, final loss {ref[2]:.6f}, finite {math.isfinite(ref[3][-1])}, "
          f"divergent runs: {bad}", flush=True)
    from nkb_classification import runtime
    if runtime._POISON:
        ok = runtime.guards_intact()
        print(f"  poison mode: guard zones around {len(runtime._GUARDED)} buffers intact: {ok}", flush=True)
        bad += 0 if ok else 1
    sys.exit(1 if bad else 0)
