"""Per-stage cost of the 256 x 256 weight-gradient kernels: the same (Cin, Cout) at several token counts (the split count is a
function of the tile count only, so time = fixed + stages_per_workgroup * c); kernel and reduce times from the library profiler."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
print("NKB_WGRAD8P", os.environ.get("NKB_WGRAD8P", "default"))
for (K, N) in [(768, 3072), (1024, 4096)]:
    for M in [64 * 7 * 8, 64 * 7 * 28, 64 * 7 * 56, 64 * 7 * 112, 64 * 7 * 224]:
        x = torch.randn(M, K, device=dev).to(T); dy = torch.randn(M, N, device=dev).to(T)
        dw = torch.zeros(N, K, device=dev)
        need = hip.conv_wgrad_workspace(d, N=M, P=1, Q=1, Cin=K, Cout=N)
        work = torch.empty(max(need, 1), device=dev)
        run = lambda: hip.conv_wgrad(d, dy, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N, workspace=work)
        for _ in range(3): run()
        torch.cuda.synchronize()
        hip.prof_enable(True)
        for _ in range(10): run()
        torch.cuda.synchronize()
        hip.prof_enable(False)
        pr = hip.prof_collect()
        kern = pr["conv_wgrad"]["ms"] / 10 * 1e3
        red = pr.get("wgrad_reduce", {"ms": 0.0})["ms"] / 10 * 1e3
        print(f"Cin={K} Cout={N} M={M:6d} stages/WG~{M // 64 // 7:4d}: kernel {kern:7.1f} us  reduce {red:5.1f} us  {2*M*K*N/kern/1e6:6.0f} TF/s  slab {need*4/1e6:.0f} MB", flush=True)
