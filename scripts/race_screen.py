"""Repeat-run screen for the kernels whose synchronisation is hand-counted (LDS-DMA weight gradient, fused attention): the
same inputs N times in one process; every result must agree with the first within accumulation-order noise."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(0)
M, K, C = 50432, 768, 2304
x = torch.randn(M, K, device=dev).to(torch.bfloat16); dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
ref = None
work = torch.empty(hip.conv_wgrad_workspace(hip.BF16, N=M, P=1, Q=1, Cin=K, Cout=C, has_bias=True), device=dev)
for it in range(N):
    dw = torch.zeros(C, K, device=dev); db = torch.zeros(C, device=dev)
    hip.conv_wgrad(hip.BF16, dy, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=C, lddy=C, R=1, S=1, stride=1, pad=0, dbias=db,
                   workspace=work)
    if ref is None: ref = (dw.clone(), db.clone())
    else: assert torch.equal(dw, ref[0]) and torch.equal(db, ref[1]), f"wgrad256 run {it} differs"
torch.cuda.synchronize()
print(f"wgrad256 (slabs + ordered reduce) x{N}: bit-identical")
# the 128 x 128 kernel on a ResNet-50 3x3 shape, same screen
Bn, Hh, Cc = 64, 28, 128
xc = torch.randn(Bn, Hh, Hh, Cc, device=dev).to(torch.bfloat16); dyc = torch.randn(Bn, Hh, Hh, Cc, device=dev).to(torch.bfloat16)
workc = torch.empty(hip.conv_wgrad_workspace(hip.BF16, N=Bn, P=Hh, Q=Hh, Cin=Cc, Cout=Cc, R=3, S=3, stride=1, pad=1), device=dev)
ref = None
for it in range(N):
    dw = torch.zeros(Cc, 3, 3, Cc, device=dev)
    hip.conv_wgrad(hip.BF16, dyc, xc, dw, N=Bn, H=Hh, W=Hh, Cin=Cc, ldx=Cc, P=Hh, Q=Hh, Cout=Cc, lddy=Cc, R=3, S=3, stride=1, pad=1,
                   workspace=workc)
    if ref is None: ref = dw.clone()
    else: assert torch.equal(dw, ref), f"conv_wgrad run {it} differs"
torch.cuda.synchronize()
print(f"conv_wgrad 3x3 (slabs + ordered reduce) x{N}: bit-identical")
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
