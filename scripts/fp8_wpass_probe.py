import os, sys, torch, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"
n_w, N, K = 96, 1024, 3072
src = [torch.randn(N * K, device=dev).to(torch.bfloat16) for _ in range(n_w)]
dst = [torch.empty(N * K, device=dev, dtype=torch.uint8) for _ in range(n_w)]
st = [torch.tensor([1., 1., 0.], device=dev) for _ in range(n_w)]
rows, nb = [], 0
for i in range(n_w):
    rows.append([src[i].data_ptr(), dst[i].data_ptr(), N * K, st[i].data_ptr(), 0, nb]); nb += hip.fp8_job_blocks(N * K)
jobs = torch.tensor(rows, dtype=torch.int64, device=dev)
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / n * 1e6
gb = n_w * N * K * 2 / 1e9
for ps in (3, 0, 2, 1):
    us = t(lambda: hip.fp8_multi(ps, jobs, n_w, nb if ps < 2 else 0))
    print(f"pass {ps}: {us:8.1f} us  ({gb / us * 1e3:.2f} TB/s on the bf16 source)" )
