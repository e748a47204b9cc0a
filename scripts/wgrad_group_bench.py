"""One transformer block's four Linear weight gradients: four nkb_conv_wgrad launches vs one nkb_wgrad_group launch (NKB_WGROUP_SPLITS
sweeps the split count; one process per value)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("NKB_WGROUP_SPLITS", os.environ.get("NKB_WGROUP_SPLITS", "model"))
for name, M, shapes in [("ViT-B/16", 50432, [(768, 2304), (768, 768), (768, 3072), (3072, 768)]),
                        ("ViT-L/14", 32768, [(1024, 3072), (1024, 1024), (1024, 4096), (4096, 1024)])]:
    jobs, singles = [], []
    for (K, N) in shapes:
        x = torch.randn(M, K, device=dev).to(T); g = torch.randn(M, N, device=dev).to(T)
        dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
        jobs.append(dict(dy=g, x=x, dw=dw, dbias=db, Cin=K, ldx=K, Cout=N, lddy=N))
        need = hip.conv_wgrad_workspace(d, N=M, P=1, Q=1, Cin=K, Cout=N, has_bias=True)
        singles.append((g, x, dw, db, K, N, torch.empty(need, device=dev)))
    def sep():
        for (g, x, dw, db, K, N, w) in singles:
            hip.conv_wgrad(d, g, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N, dbias=db, workspace=w)
    arr = hip.wgrad_jobs(jobs)
    work = torch.empty(hip.wgrad_group_workspace(False, arr, M), device=dev)
    grp = lambda: hip.wgrad_group(False, arr, M, work)
    ts, tg = min(timeit(sep) for _ in range(3)), min(timeit(grp) for _ in range(3))
    fl = sum(2.0 * M * K * N for (K, N) in shapes) / 1e6
    print(f"{name}: four launches {ts:7.1f} us ({fl / ts:5.0f} TF/s) | grouped {tg:7.1f} us ({fl / tg:5.0f} TF/s)  slabs {work.numel() * 4 / 1e6:.0f} MB", flush=True)
