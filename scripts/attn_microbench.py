"""Fused attention forward / backward-dS standalone timing (ViT-B/16: B=256, T=197, H=12, dh=64; unicom L/14: B=128, T=256, H=16)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (B, T, H) in [(256, 197, 12), (128, 256, 16)]:
    dh = 64; D = H * dh
    qkv = (torch.randn(B * T, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    o = torch.empty(B * T, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(B * H, T, device=dev)
    do = torch.randn(B * T, D, device=dev).to(torch.bfloat16)
    Tp = (T + 63) // 64 * 64
    P = torch.zeros(B * H, T, Tp, device=dev, dtype=torch.bfloat16); dS = torch.zeros_like(P); dqb = torch.empty_like(qkv)
    tf = timeit(lambda: hip.attn_forward(hip.BF16, qkv, o, lse, B, T, H, dh, dh ** -0.5))
    tb = timeit(lambda: hip.attn_backward_ds(hip.BF16, qkv, do, lse, P, dS, Tp, B, T, H, dh, dh ** -0.5, dq=dqb, ld_dq=3 * D))
    tbf = timeit(lambda: hip.attn_backward(hip.BF16, qkv, do, o, lse, dqb, B, T, H, dh, dh ** -0.5))
    gf = 4.0 * B * H * T * T * dh / 1e9
    mb = (qkv.numel() + o.numel()) * 2 / 1e6
    print(f"B={B} T={T} H={H}: fwd {tf:7.1f} us ({gf / tf * 1e3:6.1f} TF/s, {mb / tf:5.2f} TB/s of qkv+o) | bwd_ds+dQ {tb:7.1f} us | fused bwd {tbf:7.1f} us")
