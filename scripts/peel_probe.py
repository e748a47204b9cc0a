"""The ragged last row block of unicom ViT-L/14's Linears (M = 128 x 257 = 32 896 = 128.5 x 256): the launch with its ragged rows on the
companion kernel (nkb_gemm8p_ragged 1) against every row block on the persistent kernel (0) and against M = 32 768 (no ragged rows).
python scripts/peel_probe.py"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (K, N) in [(1024, 3072), (1024, 1024), (1024, 4096), (4096, 1024), (3072, 1024)]:
    row = []
    for M, rag in ((32896, False), (32896, True), (32768, True)):
        x = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) * 0.03).to(T); b = torch.zeros(N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=T)
        f = lambda: hip.conv_gemm(d, 0, x, w, y, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N, bias=b)
        hip.gemm8p_ragged(rag)
        row.append(timeit(f))
    hip.gemm8p_ragged(True)
    print(f"K={K:5d} N={N:5d}: M=32896 all on the persistent kernel {row[0]:7.1f} us | ragged rows on the companion {row[1]:7.1f} us | M=32768 {row[2]:7.1f} us", flush=True)
