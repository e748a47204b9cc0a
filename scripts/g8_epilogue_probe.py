"""Where the eight-phase GEMM's per-tile time goes: the same launches with NKB_G8_DBG = 0 (product), 1 (epilogue without its
stores), 2 (no epilogue); run once per setting (the knob is read at launch)."""
import os, sys, statistics, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
DBGS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2]
def once(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, K, N) in [(32768, 1024, 4096), (32768, 2048, 4096), (32768, 4096, 4096), (32768, 4096, 1024), (50432, 768, 3072)]:
    x = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) * 0.05).to(T)
    y = torch.empty(M, N, device=dev, dtype=T)
    run = lambda: hip.conv_gemm(d, 0, x, w, y, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N)
    out = []
    for dbg in DBGS:
        os.environ["NKB_G8_DBG"] = str(dbg)
        run(); ts = [once(run) for _ in range(5)]
        out.append(statistics.median(ts))
    f = 2.0 * M * K * N / 1e6
    print(f"M={M} K={K} N={N}: " + " | ".join(f"dbg {g}: {t:7.1f} us ({f / t:5.0f} TF/s)" for g, t in zip(DBGS, out)), flush=True)
