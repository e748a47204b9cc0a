"""ViT / unicom / ResNet 1x1 weight-gradient shapes through nkb_conv_wgrad (workspace form), one process per NKB_WGRAD8P value."""
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
print("NKB_WGRAD8P", os.environ.get("NKB_WGRAD8P", "default"))
for (M, K, N) in [(50432, 768, 2304), (50432, 768, 768), (50432, 768, 3072), (50432, 3072, 768), (32768, 1024, 3072), (32768, 1024, 4096),
                  (32768, 4096, 1024), (32768, 1024, 1024), (50176, 1024, 256), (50176, 256, 1024), (12544, 2048, 512)]:
    x = torch.randn(M, K, device=dev).to(T); dy = torch.randn(M, N, device=dev).to(T)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    need = hip.conv_wgrad_workspace(d, N=M, P=1, Q=1, Cin=K, Cout=N, R=1, S=1, stride=1, pad=0, has_bias=True)
    work = torch.empty(max(need, 1), device=dev)
    run = lambda: hip.conv_wgrad(d, dy, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N, R=1, S=1, stride=1, pad=0, dbias=db, workspace=work)
    dw.zero_(); db.zero_(); run(); torch.cuda.synchronize()
    ref = dy[:4096].float().t() @ x[:4096].float() if False else None
    t = min(timeit(run) for _ in range(3))
    # checksum against torch on a slice of the output (full product in fp32 on the GPU)
    dw.zero_(); db.zero_(); run(); torch.cuda.synchronize()
    want = (dy.float().t()[:256] @ x.float())
    err = ((dw[:256] - want).norm() / want.norm()).item()
    berr = ((db - dy.float().sum(0)).norm() / dy.float().sum(0).norm()).item()
    print(f"M={M:6d} Cin={K:5d} Cout={N:5d}: {t:7.1f} us {2*M*K*N/t/1e6:7.1f} TF/s   relerr {err:.1e} bias {berr:.1e}", flush=True)
