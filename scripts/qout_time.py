"""fp8 GEMM with and without the quantised second output (same shape, same epilogue kind): what the QOUT kernels' k-loop costs.
python scripts/qout_time.py"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
M = 32768
for (K, N) in [(1024, 4096), (4096, 1024), (1024, 3072), (1024, 1024)]:
    xq = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev); wq = torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev)
    sx = torch.tensor([1., 1e-3, 0.], device=dev); sw = torch.tensor([1., 1e-3, 0.], device=dev)
    bias = torch.zeros(N, device=dev); u6 = (torch.randn(M, N, device=dev) * 4).clamp(0, 6).to(torch.bfloat16)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16); yq = torch.empty(M, N, device=dev, dtype=torch.uint8)
    st = torch.tensor([3.0, 1.0 / 3.0, 0.0], device=dev); mask = torch.empty(M, N // 8, device=dev, dtype=torch.uint8)
    res = []
    for kind, kw in (("relu6+bias", dict(relu=2, bias=bias)), ("aux mask", dict(aux=u6, aux_mode=1)), ("plain", dict())):
        t0 = timeit(lambda: hip.gemm_fp8(0, xq, wq, y, M, K, N, deq_x=sx[1:2], deq_w=sw[1:2], **kw))
        t1 = timeit(lambda: hip.gemm_fp8(0, xq, wq, y, M, K, N, deq_x=sx[1:2], deq_w=sw[1:2], yq=yq, q_state=st, q_kind=hip.E4M3, **kw))
        res.append(f"{kind}: {t0:6.1f} -> with yq {t1:6.1f} us")
    print(f"K={K} N={N}: " + " | ".join(res), flush=True)
