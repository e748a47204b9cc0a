"""Five launches of the fused attention backward (for rocprofv3 counter passes): python one_attn.py <B> <T> <H>"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
B, T, H = [int(v) for v in sys.argv[1:4]]
dev = "cuda"; dh = 64; D = H * dh
qkv = (torch.randn(B * T, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
o = torch.empty(B * T, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(B * H, T, device=dev)
do = torch.randn(B * T, D, device=dev).to(torch.bfloat16); dqb = torch.empty_like(qkv)
hip.attn_forward(hip.BF16, qkv, o, lse, B, T, H, dh, dh ** -0.5)
for _ in range(5):
    hip.attn_backward(hip.BF16, qkv, do, o, lse, dqb, B, T, H, dh, dh ** -0.5)
torch.cuda.synchronize()
