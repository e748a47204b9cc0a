import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
ci, co, k, s, h, mode = [int(v) for v in sys.argv[1:7]]
B = 256; pad = k // 2; P = (h + 2 * pad - k) // s + 1
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
x = torch.randn(B, h, h, ci, device=dev).to(T); w = torch.randn(co, k, k, ci, device=dev).to(T) * 0.05
y = torch.empty(B, P, P, co, device=dev, dtype=T)
for _ in range(5):
    hip.conv_gemm(d, 0, x, w, y, N=B, H=h, W=h, Cin=ci, ldx=ci, P=P, Q=P, Cout=co, ldy=co, R=k, S=k, stride=s, pad=pad)
torch.cuda.synchronize()
