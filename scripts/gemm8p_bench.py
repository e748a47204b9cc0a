"""Same-process A/B of the eight-phase GEMM core against the 128 x 128 kernel (and torch.mm as a yardstick) on the ViT-B/16,
unicom ViT-L/14 and ResNet-50 layer3/4 GEMM shapes; interleaved rounds, median and min (guide rule 24)."""
import os, sys, statistics, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
SHAPES = [(50432, 768, 2304), (50432, 768, 768), (50432, 768, 3072), (50432, 3072, 768), (50432, 2304, 768),
          (32768, 1024, 3072), (32768, 1024, 4096), (32768, 4096, 1024), (32768, 1024, 1024),
          (50176, 1024, 256), (50176, 256, 1024), (12544, 2048, 512), (12544, 512, 2048), (200704, 512, 256), (50176, 1024, 512)]
def once(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, K, N) in SHAPES:
    x = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) * 0.05).to(T)
    y = torch.empty(M, N, device=dev, dtype=T)
    run = lambda: hip.conv_gemm(d, 0, x, w, y, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N)
    mm = lambda: torch.mm(x, w.t(), out=y)
    res = {"g8": [], "k128": [], "mm": []}
    for rnd in range(5):
        hip.gemm8p_config(True, 1, 128); run(); res["g8"].append(once(run))
        hip.gemm8p_config(False); run(); res["k128"].append(once(run))
        mm(); res["mm"].append(once(mm))
    f = 2.0 * M * K * N / 1e6
    line = f"M={M:6d} K={K:5d} N={N:5d}:"
    for k in ("g8", "k128", "mm"):
        med, mn = statistics.median(res[k]), min(res[k])
        line += f"  {k} {med:7.1f} us ({f / med:6.1f} TF/s, best {f / mn:6.1f})"
    print(line, flush=True)
