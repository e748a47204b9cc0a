"""Repeat-run screen for the kernels whose synchronisation is hand-counted (LDS-DMA weight gradient, fused attention): the
same inputs N times in one process; every result must agree with the first within accumulation-order noise."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(0)
M, K, C = 50432, 768, 2304
x = torch.randn(M, K, device=dev).to(torch.bfloat16); dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
ref = None; worst = 0.0
for it in range(N):
    dw = torch.zeros(C, K, device=dev); db = torch.zeros(C, device=dev)
    hip.conv_wgrad(hip.BF16, dy, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=C, lddy=C, R=1, S=1, stride=1, pad=0, dbias=db)
    if ref is None: ref = (dw.clone(), db.clone()); scale = dw.abs().max().item()
    else: worst = max(worst, (dw - ref[0]).abs().max().item() / scale, (db - ref[1]).abs().max().item() / ref[1].abs().max().item())
torch.cuda.synchronize()
print(f"wgrad256 x{N}: worst relative deviation from run 0 = {worst:.2e}")
assert worst < 1e-4
B, T, H, dh = 64, 197, 12, 64; D = H * dh
qkv = (torch.randn(B * T, 3 * D, device=dev) * 0.5).to(torch.bfloat16); do = torch.randn(B * T, D, device=dev).to(torch.bfloat16)
first = None
for it in range(N):
    o = torch.empty(B * T, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(B * H, T, device=dev)
    dq = torch.empty_like(qkv)
    hip.attn_forward(hip.BF16, qkv, o, lse, B, T, H, dh, dh ** -0.5)
    hip.attn_backward(hip.BF16, qkv, do, o, lse, dq, B, T, H, dh, dh ** -0.5)
    if first is None: first = (o.clone(), lse.clone(), dq.clone())
    else:
        assert torch.equal(o, first[0]) and torch.equal(lse, first[1]) and torch.equal(dq, first[2]), f"attention run {it} differs"
torch.cuda.synchronize()
print(f"attention fwd + fused bwd x{N}: bit-identical")
