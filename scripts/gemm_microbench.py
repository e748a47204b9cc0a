"""nkb_conv_gemm as a plain GEMM (R=S=1) on ViT-B/16 and ResNet-50 1x1 shapes, next to torch.mm (yardstick only)."""
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
for (M, K, N) in [(50432, 768, 2304), (50432, 768, 768), (50432, 768, 3072), (50432, 3072, 768), (50432, 2304, 768),
                  (50176, 1024, 256), (50176, 256, 1024), (200704, 512, 256), (12544, 2048, 512), (12544, 512, 2048)]:
    x = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) * 0.05).to(T)
    y = torch.empty(M, N, device=dev, dtype=T)
    t = timeit(lambda: hip.conv_gemm(d, 0, x, w, y, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N))
    tv = timeit(lambda: torch.mm(x, w.t(), out=y))
    print(f"M={M:6d} K={K:5d} N={N:5d}: nkb {t:7.1f} us {2*M*K*N/t/1e6:7.1f} TF/s | torch.mm {tv:7.1f} us {2*M*K*N/tv/1e6:7.1f} TF/s")
