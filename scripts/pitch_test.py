"""Does a power-of-two activation row pitch cost L2 channel conflicts?  1x1 conv fwd with ldx = Cin vs Cin + pad."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, K, N) in [(50176, 1024, 256), (50176, 256, 1024), (12544, 2048, 512), (12544, 512, 2048), (200704, 512, 128), (200704, 512, 256), (50432, 768, 768), (50432, 3072, 768)]:
    w = (torch.randn(N, K, device=dev) * 0.05).to(T)
    res = []
    for padx, pady in [(0, 0), (64, 0), (0, 64), (64, 64), (8, 8)]:
        x = torch.randn(M, K + padx, device=dev).to(T)
        y = torch.empty(M, N + pady, device=dev, dtype=T)
        t = timeit(lambda: hip.conv_gemm(d, 0, x, w, y, N=M, H=1, W=1, Cin=K, ldx=K + padx, P=1, Q=1, Cout=N, ldy=N + pady))
        res.append(t)
    print(f"M={M:6d} K={K:5d} N={N:5d}: pitch K,N {res[0]:6.1f} | x+64 {res[1]:6.1f} | y+64 {res[2]:6.1f} | both+64 {res[3]:6.1f} | both+8 {res[4]:6.1f} us")
