"""How long do the last M % 256 rows of a unicom ViT-L/14 Linear take on the 128 x 128 tile kernel (the eight-phase kernel switched off)?
python scripts/peel_probe.py [rows]"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (K, N) in [(1024, 3072), (1024, 1024), (1024, 4096), (4096, 1024), (3072, 1024), (768, 768), (3072, 768)]:
    for M in (rows, 32896, 32768):
        x = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) * 0.03).to(T); b = torch.zeros(N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=T)
        f = lambda: hip.conv_gemm(d, 0, x, w, y, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N, bias=b)
        hip.gemm8p_config(M > 1000, 1, 128)
        t = timeit(f)
        print(f"M={M:6d} K={K:5d} N={N:5d}: {t:8.1f} us  ({2.0 * M * K * N / t / 1e6:7.1f} TF/s)", flush=True)
