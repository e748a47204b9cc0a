import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
M, K, N = 50432, 768, 3072
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
x = torch.randn(M, K, device=dev).to(T); dy = torch.randn(M, N, device=dev).to(T)
dw = torch.zeros(N, K, device=dev)
for _ in range(3):
    hip.conv_wgrad(d, dy, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N)
torch.cuda.synchronize()
