"""one fp8 GEMM with the quantised second output (the launch that aborted): python scripts/qout_probe.py M K N [relu_bias|aux|add]"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
M, K, N = [int(v) for v in sys.argv[1:4]]; kind = sys.argv[4] if len(sys.argv) > 4 else "relu_bias"
dev = "cuda"
x = torch.randn(M, K); w = torch.randn(N, K) * 0.05
xq = x.to(torch.float8_e4m3fn).view(torch.uint8).to(dev); wq = w.to(torch.float8_e4m3fn).view(torch.uint8).to(dev)
sx = torch.tensor([1., 1., 0.], device=dev); sw = torch.tensor([1., 1., 0.], device=dev)
bias = torch.randn(N).to(dev); add = torch.randn(M, N).to(torch.bfloat16).to(dev); u6 = (torch.randn(M, N) * 4).clamp(0, 6).to(torch.bfloat16).to(dev)
kw = dict(relu_bias=dict(relu=2, bias=bias), aux=dict(aux=u6, aux_mode=1), add=dict(add=add, ldadd=N), plain=dict())[kind]
y = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
st = torch.tensor([3.0, 1.0 / 3.0, 0.0], device=dev); yq = torch.full((M, N), 0x55, device=dev, dtype=torch.uint8)
print("launch", kind, flush=True)
hip.gemm_fp8(0, xq, wq, y, M, K, N, deq_x=sx[1:2], deq_w=sw[1:2], yq=yq, q_state=st, q_kind=hip.E4M3, **kw)
torch.cuda.synchronize()
print("ok", float(y.float().abs().max()), st.tolist(), flush=True)
