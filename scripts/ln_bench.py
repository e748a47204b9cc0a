"""LayerNorm backward (+ its parameter-gradient reduce) at the ViT-B/16 and unicom ViT-L/14 sizes, kernel times from the library profiler."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
print("NKB_LN_BWD_BLOCKS", os.environ.get("NKB_LN_BWD_BLOCKS", "default"))
for (M, D) in [(50432, 768), (32768, 1024)]:
    x = torch.randn(M, D, device=dev).to(T); g = torch.randn(M, D, device=dev).to(T); add = torch.randn(M, D, device=dev).to(T)
    gamma = torch.randn(D, device=dev); beta = torch.randn(D, device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
    work = torch.empty(hip.layernorm_ws(D), device=dev)
    fwd = lambda: hip.layernorm_fwd(d, x, D, gamma, beta, y, D, mean, rstd, M, D, 1e-6)
    bwd = lambda: hip.layernorm_bwd(d, g, D, x, D, gamma, mean, rstd, add, dx, D, dg, db, M, D, workspace=work)
    fwd()
    cs = torch.zeros(D, device=dev); st = torch.tensor([100.0, 0.01, 0.0], device=dev); yq = torch.empty(M, D, device=dev, dtype=torch.uint8)
    rsc = torch.ones(M // 256 + 1, device=dev)
    bwdq = lambda: hip.layernorm_bwd(d, g, D, x, D, gamma, mean, rstd, add, dx, D, dg, db, M, D, workspace=work, yq=yq, q_state=st,
                                     q_kind=hip.E5M2, row_scale=rsc, rows_per_sample=256, colsum=cs)
    cases = [(fwd, "fwd"), (bwd, "bwd+param")] + ([(bwdq, "bwd+param+fp8")] if D % 256 == 0 else [])
    for fn, name in cases:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        hip.prof_enable(True)
        for _ in range(10): fn()
        torch.cuda.synchronize()
        hip.prof_enable(False)
        pr = hip.prof_collect()
        print(f"M={M} D={D} {name}: " + "  ".join(f"{k} {v['ms'] / 10 * 1e3:6.1f} us" for k, v in pr.items() if v["ms"] > 0), flush=True)
