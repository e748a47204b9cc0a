"""One 1x1 weight-gradient shape through nkb_conv_wgrad a few times (for rocprofv3 --pmc passes: scripts/pmc_one.sh <tag> wgradr one_wr.py M Cin Cout)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
M, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (50176, 256, 1024)
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
x = torch.randn(M, K, device=dev).to(T); dy = torch.randn(M, N, device=dev).to(T)
dw = torch.zeros(N, K, device=dev)
ws = torch.empty(hip.conv_wgrad_workspace(d, N=M, P=1, Q=1, Cin=K, Cout=N), device=dev)
for _ in range(5):
    hip.conv_wgrad(d, dy, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N, workspace=ws)
torch.cuda.synchronize()
