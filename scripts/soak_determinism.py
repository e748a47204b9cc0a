"""Soak of the real ResNet-50 bs-256 bf16 step (recorded plans replayed in C, weight gradients co-running on the second stream): two
runs of N steps from the same seed must leave bit-identical parameters and BatchNorm running statistics — the step has no atomics
in its arithmetic, so any difference is a race or a stale read.  (Round 3: a one-launch form of the BatchNorm tile statistics passed
every op-level bit-identity test and failed THIS after 150 steps — its last block read partials of the previous launch from its
XCD's L2.)  Usage: python scripts/soak_determinism.py [steps] [model]"""
import argparse, os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
import bench


def run(steps, model_name="resnet50", batch=256, dtype="bf16", heads=""):
    args = argparse.Namespace(model=model_name, classes=1000, batch=batch, dtype=dtype, heads=heads)
    dev = torch.device("cuda:0")
    model, opt, crit = bench.build(args, dev)
    model.fp8_linear = dtype == "fp8"
    g = torch.Generator().manual_seed(7)
    img = torch.randn(batch, 3, 224, 224, generator=g).to(dev)
    tgt = torch.randint(0, 1000, (batch,), generator=g).to(dev)
    model.train()
    torch.manual_seed(99)                      # stochastic-depth / dropout seed stream
    for _ in range(steps):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(model(img), tgt)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    bufs = [b.detach().float().flatten() for b in model.buffers() if b.is_floating_point()]
    return model.arena.flat_param.clone(), (torch.cat(bufs).clone() if bufs else torch.zeros(1)), loss.item()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    name = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
    a, b = run(n, name, 256 if "unicom" not in name else 128), run(n, name, 256 if "unicom" not in name else 128)
    same = torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    print(f"{name}: {n} steps twice, final losses {a[2]:.6f} / {b[2]:.6f}, bit-identical: {same}")
    sys.exit(0 if same else 1)
