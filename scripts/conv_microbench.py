"""Times every distinct ResNet-50 conv shape (bs 256, bf16) through nkb_conv_gemm fwd / dgrad and nkb_conv_wgrad,
with the ring kernel off and on.  Prints us, TFLOP/s and achieved GB/s vs the tensor-byte floor."""
import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [  # Cin, Cout, k, s, Hin
 (64,64,1,1,56),(64,64,3,1,56),(64,256,1,1,56),(256,64,1,1,56),(256,128,1,1,56),(128,128,3,2,56),(128,512,1,1,28),
 (256,512,1,2,56),(512,128,1,1,28),(128,128,3,1,28),(512,256,1,1,28),(256,256,3,2,28),(256,1024,1,1,14),(512,1024,1,2,28),
 (1024,256,1,1,14),(256,256,3,1,14),(1024,512,1,1,14),(512,512,3,2,14),(512,2048,1,1,7),(1024,2048,1,2,14),(2048,512,1,1,7),(512,512,3,1,7)]
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print(f"{'shape':28s} {'GF':>7s} {'MBmin':>6s} | {'fwd0':>7s} {'fwd1':>7s} | {'dgr0':>7s} {'dgr1':>7s} | {'wgrad':>7s}   (us; 0=staged,1=ring)")
tot = [0, 0, 0, 0, 0]
for (ci, co, k, s, h) in SHAPES:
    pad = k // 2; P = (h + 2 * pad - k) // s + 1
    x = torch.randn(B, h, h, ci, device=dev).to(T); w = torch.randn(co, k, k, ci, device=dev).to(T) * 0.05
    wt = w.permute(3, 1, 2, 0).contiguous(); y = torch.empty(B, P, P, co, device=dev, dtype=T); dx = torch.empty_like(x)
    dw = torch.zeros(co, k, k, ci, device=dev)
    stats = torch.empty(hip.bn_stats_floats(max(hip.stat_tiles(d, B * P * P, co), (B * P * P + 127) // 128 * 8), co), device=dev)
    fwd = lambda: hip.conv_gemm(d, 0, x, w, y, N=B, H=h, W=h, Cin=ci, ldx=ci, P=P, Q=P, Cout=co, ldy=co, R=k, S=k, stride=s, pad=pad, stats=stats)
    dgr = lambda: hip.conv_gemm(d, 1, y, wt, dx, N=B, H=P, W=P, Cin=co, ldx=co, P=h, Q=h, Cout=ci, ldy=ci, R=k, S=k, stride=s, pad=pad)
    wgr = lambda: hip.conv_wgrad(d, y, x, dw, N=B, H=h, W=h, Cin=ci, ldx=ci, P=P, Q=P, Cout=co, lddy=co, R=k, S=k, stride=s, pad=pad)
    res = []
    for mode in (0, 1):
        hip.load().nkb_set_ring(mode)
        res.append((timeit(fwd), timeit(dgr)))
    tw = timeit(wgr)
    gf = 2 * B * P * P * co * ci * k * k / 1e9
    mb = (x.numel() + y.numel()) * 2 / 1e6
    print(f"{ci:4d}->{co:4d} k{k} s{s} {h:3d}->{P:3d}  {gf:7.1f} {mb:6.0f} | {res[0][0]:7.1f} {res[1][0]:7.1f} | {res[0][1]:7.1f} {res[1][1]:7.1f} | {tw:7.1f}")
    for i, v in enumerate((res[0][0], res[1][0], res[0][1], res[1][1], tw)): tot[i] += v
print("sum of distinct shapes (us):", [round(v) for v in tot])
