"""Yardstick only (not used by the product): what the vendor GEMM (torch.mm -> hipBLASLt) reaches on the 1x1-conv shapes."""
import torch
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, K, N) in [(802816, 64, 256), (802816, 256, 64), (802816, 64, 64), (802816, 256, 128), (200704, 128, 512), (200704, 512, 128),
                  (50176, 256, 1024), (50176, 1024, 256), (12544, 512, 2048), (12544, 2048, 512), (200704, 1152, 128), (50176, 2304, 256)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: torch.mm(x, w.t(), out=y))
    mb = (M * K + M * N + N * K) * 2 / 1e6
    print(f"M={M:7d} K={K:5d} N={N:5d}: {t:7.1f} us  {2*M*K*N/t/1e6:7.1f} TF/s  {mb/t/1e3:5.2f} TB/s (floor {mb/5.0:6.1f} us)")
