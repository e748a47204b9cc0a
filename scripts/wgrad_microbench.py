"""nkb_conv_wgrad on the ViT-B/16 linear layers (1x1, reduction over 50 432 tokens) and the ResNet-50 3x3 shapes, next to
torch.mm(dy.T, x) (yardstick only).  NKB_WGRAD_WGS sweeps the split target."""
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
print("WGS", os.environ.get("NKB_WGRAD_WGS", "default"))
for (M, K, N) in [(50432, 768, 2304), (50432, 768, 768), (50432, 768, 3072), (50432, 3072, 768),
                  (50176, 1024, 256), (50176, 256, 1024), (200704, 512, 128), (12544, 2048, 512), (802816, 64, 256)]:
    x = torch.randn(M, K, device=dev).to(T); dy = torch.randn(M, N, device=dev).to(T)
    dw = torch.zeros(N, K, device=dev)
    t = timeit(lambda: hip.conv_wgrad(d, dy, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N, R=1, S=1, stride=1, pad=0))
    out = torch.empty(N, K, device=dev, dtype=T)
    tv = timeit(lambda: torch.mm(dy.t(), x, out=out))
    print(f"M={M:6d} Cin={K:5d} Cout={N:5d}: nkb {t:7.1f} us {2*M*K*N/t/1e6:7.1f} TF/s | torch.mm {tv:7.1f} us {2*M*K*N/tv/1e6:7.1f} TF/s")
B = 256
for (ci, co, h) in [(64, 64, 56), (128, 128, 28), (256, 256, 14), (512, 512, 7)]:
    x = torch.randn(B, h, h, ci, device=dev).to(T); y = torch.randn(B, h, h, co, device=dev).to(T)
    dw = torch.zeros(co, 3, 3, ci, device=dev)
    t = timeit(lambda: hip.conv_wgrad(d, y, x, dw, N=B, H=h, W=h, Cin=ci, ldx=ci, P=h, Q=h, Cout=co, lddy=co, R=3, S=3, stride=1, pad=1))
    print(f"3x3 {ci:4d}->{co:4d} {h:3d}x{h:<3d}: nkb {t:7.1f} us {2*B*h*h*ci*co*9/t/1e6:7.1f} TF/s")
