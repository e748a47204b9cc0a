"""Cost of the epilogue variants on the write-heavy 1x1 shapes: no stats / stats / residual add."""
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
for (M, K, N) in [(802816, 64, 256), (802816, 64, 64), (802816, 256, 64), (200704, 128, 512), (50176, 256, 1024), (50176, 1024, 256)]:
    x = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) * 0.05).to(T)
    y = torch.empty(M, N, device=dev, dtype=T); add = torch.randn(M, N, device=dev).to(T)
    tiles = hip.stat_tiles(d, M, N)
    stats = torch.empty(hip.bn_stats_floats(tiles, N), device=dev)
    g = dict(N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N)
    t0 = timeit(lambda: hip.conv_gemm(d, 0, x, w, y, **g))
    t1 = timeit(lambda: hip.conv_gemm(d, 0, x, w, y, stats=stats, **g))
    t2 = timeit(lambda: hip.conv_gemm(d, 0, x, w, y, add=add, ldadd=N, **g))
    mb = (M * K + M * N) * 2 / 1e6
    print(f"M={M:6d} K={K:4d} N={N:4d}: plain {t0:6.1f} us ({mb/t0/1e3:4.2f} TB/s) | +stats {t1:6.1f} | +add {t2:6.1f}")
