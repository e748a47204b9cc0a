"""Every 1x1 weight-gradient shape of the ResNet-50 bs-256 step through nkb_conv_wgrad (deterministic slabs, as the engine calls
it), alone on the GPU: us per launch, which kernel took it, the HBM floor (operands once at 5.5 TB/s) and the MFMA floor (1.2 PFLOP/s)."""
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
SHAPES = [("l1 conv1 b0", 802816, 64, 64), ("l1 conv1", 802816, 256, 64), ("l1 R / ds", 802816, 64, 256),
          ("l2 conv1 b0", 802816, 256, 128), ("l2 conv1", 200704, 512, 128), ("l2 R", 200704, 128, 512), ("l2 ds", 200704, 256, 512),
          ("l3 conv1 b0", 200704, 512, 256), ("l3 conv1", 50176, 1024, 256), ("l3 conv3", 50176, 256, 1024), ("l3 ds", 50176, 512, 1024),
          ("l4 conv1 b0", 50176, 1024, 512), ("l4 conv1", 12544, 2048, 512), ("l4 conv3", 12544, 512, 2048), ("l4 ds", 12544, 1024, 2048)]
only = sys.argv[1:]
for name, M, K, N in SHAPES:
    if only and not any(o in name for o in only): continue
    x = torch.randn(M, K, device=dev).to(T); dy = torch.randn(M, N, device=dev).to(T)
    dw = torch.zeros(N, K, device=dev)
    ws = torch.empty(hip.conv_wgrad_workspace(d, N=M, P=1, Q=1, Cin=K, Cout=N), device=dev)
    n0 = hip.kernel_launches("wgrad8p")
    t = timeit(lambda: hip.conv_wgrad(d, dy, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N, workspace=ws))
    kern = "wgrad8p" if hip.kernel_launches("wgrad8p") > n0 else "generic"
    byt = 2.0 * M * (K + N)
    fl = 2.0 * M * K * N
    print(f"{name:12s} M={M:6d} Cin={K:5d} Cout={N:5d} {kern:8s} {t:7.1f} us | hbm floor {byt / 5.5e6:6.1f} us  mfma floor {fl / 1.2e9:6.1f} us"
          f" | slabs {ws.numel() * 4 / 1e6:6.1f} MB", flush=True)
