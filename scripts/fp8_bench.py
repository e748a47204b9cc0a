"""fp8 (e4m3 x e4m3) vs bf16 on the eight-phase GEMM core, unicom ViT-L/14 and ViT-B/16 shapes, same process, interleaved."""
import os, sys, statistics, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16
def once(fn, n=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (M, K, N) in [(32768, 1024, 3072), (32768, 1024, 4096), (32768, 4096, 1024), (32768, 1024, 1024), (50432, 768, 2304), (50432, 768, 3072), (50432, 3072, 768)]:
    x = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) * 0.05).to(T)
    y = torch.empty(M, N, device=dev, dtype=T)
    sx, sw = torch.tensor([1., 1., 0.], device=dev), torch.tensor([1., 1., 0.], device=dev)
    hip.fp8_amax(hip.BF16, x, x.numel(), sx); hip.fp8_scale_update(sx, 0)
    hip.fp8_amax(hip.BF16, w, w.numel(), sw); hip.fp8_scale_update(sw, 0)
    xq = torch.empty(M, K, device=dev, dtype=torch.uint8); wq = torch.empty(N, K, device=dev, dtype=torch.uint8)
    quant = lambda: hip.fp8_quantize(hip.BF16, 0, x, x.numel(), sx, xq)
    quant(); hip.fp8_quantize(hip.BF16, 0, w, w.numel(), sw, wq)
    f8 = lambda: hip.gemm_fp8(0, xq, wq, y, M, K, N, deq_x=sx[1:2], deq_w=sw[1:2])
    b16 = lambda: hip.conv_gemm(hip.BF16, 0, x, w, y, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N)
    res = {"fp8": [], "bf16": [], "quant": []}
    for rnd in range(5):
        f8(); res["fp8"].append(once(f8)); b16(); res["bf16"].append(once(b16)); quant(); res["quant"].append(once(quant))
    f = 2.0 * M * K * N / 1e6
    print(f"M={M:6d} K={K:5d} N={N:5d}: fp8 {statistics.median(res['fp8']):7.1f} us ({f / statistics.median(res['fp8']):6.0f} TF/s)  "
          f"bf16 {statistics.median(res['bf16']):7.1f} us ({f / statistics.median(res['bf16']):6.0f} TF/s)  quantize x {statistics.median(res['quant']):6.1f} us", flush=True)
