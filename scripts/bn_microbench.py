"""Achieved HBM bandwidth of the BatchNorm kernels on the ResNet-50 (bs 256) stage shapes.
Usage (GPU box): python scripts/bn_microbench.py [--dtype bf16]"""
import os
import sys
import argparse

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nkb-classification_amd"))
from nkb_classification import hip  # noqa: E402

SHAPES = [  # rows, C, residual stage?
    (802816, 64, False), (802816, 256, True), (802816, 128, False), (200704, 128, False), (200704, 512, True),
    (50176, 256, False), (50176, 1024, True), (12544, 512, False), (12544, 2048, True),
]


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    T = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    d = hip.dt(T)
    es = 2 if T == torch.bfloat16 else 4
    dev = "cuda:0"
    tot = {"apply": 0.0, "reduce+apply": 0.0}
    print(f"{'rows':>8} {'C':>5} res | apply us  TB/s | bwd us  TB/s")
    for rows, C, res in SHAPES:
        c = torch.randn(rows, C, device=dev).to(T)
        y = torch.empty_like(c)
        r = torch.randn(rows, C, device=dev).to(T) if res else None
        g = torch.randn(rows, C, device=dev).to(T)
        gc = torch.empty_like(c)
        scale = torch.rand(C, device=dev) + 0.5
        shift = torch.randn(C, device=dev) * 0.1
        mean = torch.zeros(C, device=dev)
        invstd = torch.ones(C, device=dev)
        gamma = torch.ones(C, device=dev)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        work = torch.empty(hip.bn_backward_ws(rows, C), device=dev)
        t_apply = timeit(lambda: hip.bn_apply(d, c, r, y, scale, shift, rows, C, True))
        b_apply = rows * C * es * (3 if res else 2)
        if res:
            fn = lambda: hip.bn_backward(d, g, c, y, mean, invstd, gamma, rows, C, dg, db, gc, g, work)
            b_bwd = rows * C * es * (3 + 3 + 2)
        else:
            fn = lambda: hip.bn_backward(d, g, c, None, mean, invstd, gamma, rows, C, dg, db, gc, None, work,
                                         fscale=scale, fshift=shift)
            b_bwd = rows * C * es * (2 + 2 + 1)
        t_bwd = timeit(fn)
        if res:
            fr = lambda: hip.bn_backward(d, g, c, y, mean, invstd, gamma, rows, C, dg, db, None, None, work)
        else:
            fr = lambda: hip.bn_backward(d, g, c, None, mean, invstd, gamma, rows, C, dg, db, None, None, work,
                                         fscale=scale, fshift=shift)
        t_red = timeit(fr)
        b_red = rows * C * es * (3 if res else 2)
        tot["apply"] += t_apply
        tot["reduce+apply"] += t_bwd
        print(f"{rows:8d} {C:5d} {int(res):3d} | {t_apply:8.1f} {b_apply / t_apply / 1e6:5.2f} | {t_bwd:8.1f} {b_bwd / t_bwd / 1e6:5.2f} | reduce {t_red:7.1f} {b_red / t_red / 1e6:5.2f}")
    print("sum us:", {k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
