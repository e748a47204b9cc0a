"""Per-launch device time of the weight-gradient kernels and of their ordered reduce pass on the ResNet-50 shapes (bs 256, bf16),
from the library's own HIP-event profiler (one event pair per launch)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16; B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [(64,64,1,1,56),(64,64,3,1,56),(64,256,1,1,56),(256,64,1,1,56),(256,128,1,1,56),(128,128,3,2,56),(128,512,1,1,28),(512,128,1,1,28),
          (128,128,3,1,28),(512,256,1,1,28),(256,256,3,2,28),(256,1024,1,1,14),(1024,256,1,1,14),(256,256,3,1,14),(1024,512,1,1,14),
          (512,512,3,2,14),(512,2048,1,1,7),(2048,512,1,1,7),(512,512,3,1,7)]
print(f"{'shape':24s} {'GF':>6s} | {'kernel us':>9s} {'TF/s':>6s} | {'reduce us':>9s} | slab MB")
for (ci, co, k, s, h) in SHAPES:
    pad = k // 2; P = (h + 2 * pad - k) // s + 1
    x = torch.randn(B, h, h, ci, device=dev).to(T); y = torch.randn(B, P, P, co, device=dev).to(T)
    dw = torch.zeros(co, k, k, ci, device=dev)
    need = hip.conv_wgrad_workspace(d, N=B, P=P, Q=P, Cin=ci, Cout=co, R=k, S=k, stride=s, pad=pad)
    work = torch.empty(need, device=dev)
    run = lambda: hip.conv_wgrad(d, y, x, dw, N=B, H=h, W=h, Cin=ci, ldx=ci, P=P, Q=P, Cout=co, lddy=co, R=k, S=k, stride=s, pad=pad, workspace=work)
    for _ in range(3): run()
    torch.cuda.synchronize()
    hip.prof_enable(True)
    for _ in range(10): run()
    torch.cuda.synchronize()
    hip.prof_enable(False)
    pr = hip.prof_collect()
    kern = pr["conv_wgrad"]["ms"] / 10 * 1e3
    red = pr.get("wgrad_reduce", {"ms": 0.0})["ms"] / 10 * 1e3
    gf = 2 * B * P * P * co * ci * k * k / 1e9
    print(f"{ci:4d}->{co:4d} k{k} s{s} {h:3d}->{P:3d} {gf:6.1f} | {kern:9.1f} {gf / kern * 1e3:6.0f} | {red:9.1f} | {need * 4 / 1e6:6.1f}", flush=True)
