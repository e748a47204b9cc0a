"""Times every distinct ResNet-50 conv shape (bs 256, bf16) through nkb_conv_gemm fwd / dgrad and nkb_conv_wgrad and
compares with the HBM floor (tensor bytes / 5 TB/s) and the MFMA floor (flops / 2.5 PF/s).  x<count> = occurrences of
the shape in one ResNet-50 step; the last column is the time above the floor summed over the step."""
import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SHAPES = [  # Cin, Cout, k, s, Hin, count
 (64,64,1,1,56,1),(64,64,3,1,56,3),(64,256,1,1,56,4),(256,64,1,1,56,2),(256,128,1,1,56,1),(128,128,3,2,56,1),(128,512,1,1,28,4),
 (256,512,1,2,56,1),(512,128,1,1,28,3),(128,128,3,1,28,3),(512,256,1,1,28,1),(256,256,3,2,28,1),(256,1024,1,1,14,6),(512,1024,1,2,28,1),
 (1024,256,1,1,14,5),(256,256,3,1,14,5),(1024,512,1,1,14,1),(512,512,3,2,14,1),(512,2048,1,1,7,3),(1024,2048,1,2,14,1),(2048,512,1,1,7,2),(512,512,3,1,7,2)]
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print(f"{'shape':26s} {'cnt':>3s} {'GF':>6s} {'MB':>5s} {'floor':>6s} | {'fwd':>7s} {'dgrad':>7s} {'wgrad':>7s} | over-floor x cnt (us)")
tot = [0.0, 0.0, 0.0, 0.0]
for (ci, co, k, s, h, cnt) in SHAPES:
    pad = k // 2; P = (h + 2 * pad - k) // s + 1
    x = torch.randn(B, h, h, ci, device=dev).to(T); w = torch.randn(co, k, k, ci, device=dev).to(T) * 0.05
    wt = w.permute(3, 1, 2, 0).contiguous(); y = torch.empty(B, P, P, co, device=dev, dtype=T); dx = torch.empty_like(x)
    dw = torch.zeros(co, k, k, ci, device=dev)
    stats = torch.empty(hip.bn_stats_floats(max(hip.stat_tiles(d, B * P * P, co), (B * P * P + 127) // 128 * 8), co), device=dev)
    fwd = lambda: hip.conv_gemm(d, 0, x, w, y, N=B, H=h, W=h, Cin=ci, ldx=ci, P=P, Q=P, Cout=co, ldy=co, R=k, S=k, stride=s, pad=pad, stats=stats)
    dgr = lambda: hip.conv_gemm(d, 1, y, wt, dx, N=B, H=P, W=P, Cin=co, ldx=co, P=h, Q=h, Cout=ci, ldy=ci, R=k, S=k, stride=s, pad=pad)
    work = torch.empty(hip.conv_wgrad_workspace(d, N=B, P=P, Q=P, Cin=ci, Cout=co, R=k, S=k, stride=s, pad=pad), device=dev)
    wgr = lambda: hip.conv_wgrad(d, y, x, dw, N=B, H=h, W=h, Cin=ci, ldx=ci, P=P, Q=P, Cout=co, lddy=co, R=k, S=k, stride=s, pad=pad, workspace=work)
    tf, tg, tw = timeit(fwd), timeit(dgr), timeit(wgr)
    gf = 2 * B * P * P * co * ci * k * k / 1e9
    mb = (x.numel() + y.numel() + w.numel()) * 2 / 1e6
    floor = max(mb / 5.0, gf / 2.5)        # us: MB / (5 TB/s) and GF / (2.5 PF/s)
    over = [(t - floor) * cnt for t in (tf, tg, tw)]
    print(f"{ci:4d}->{co:4d} k{k} s{s} {h:3d}->{P:3d} {cnt:3d} {gf:6.1f} {mb:5.0f} {floor:6.1f} | {tf:7.1f} {tg:7.1f} {tw:7.1f} | {over[0]:6.0f} {over[1]:6.0f} {over[2]:6.0f}")
    tot[0] += floor * cnt
    for i, t in enumerate((tf, tg, tw)): tot[i + 1] += t * cnt
print("step totals (us): floor per pass %.0f | fwd %.0f dgrad %.0f wgrad %.0f" % tuple(tot))
