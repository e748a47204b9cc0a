"""Five launches of one plain GEMM through nkb_conv_gemm (for rocprofv3 counter passes): python one_gemm.py <M> <K> <N>"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
M, K, N = [int(v) for v in sys.argv[1:4]]
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
x = torch.randn(M, K, device=dev).to(T); w = (torch.randn(N, K, device=dev) * 0.05).to(T); b = torch.zeros(N, device=dev)
y = torch.empty(M, N, device=dev, dtype=T)
for _ in range(5):
    hip.conv_gemm(d, 0, x, w, y, N=1, H=M, W=1, Cin=K, ldx=K, P=M, Q=1, Cout=N, ldy=N, bias=b)
torch.cuda.synchronize()
