"""Where a head of the fused attention backward spends its time (diagnostic build: scripts/build_alt.sh stamps "-DNKB_ATTN_STAMPS" attention.hip,
NKBHIP_LIB=build/alt_stamps/libnkbhip.so): python scripts/attn_stamps.py B T H"""
import ctypes, os, sys, torch, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
B, T, H = [int(v) for v in sys.argv[1:4]]
dev = "cuda"; dh = 64; D = H * dh
qkv = (torch.randn(B * T, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
o = torch.empty(B * T, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(B * H, T, device=dev)
do = torch.randn(B * T, D, device=dev).to(torch.bfloat16); dqb = torch.empty_like(qkv)
hip.attn_forward(hip.BF16, qkv, o, lse, B, T, H, dh, dh ** -0.5)
for _ in range(3): hip.attn_backward(hip.BF16, qkv, do, o, lse, dqb, B, T, H, dh, dh ** -0.5)
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); hip.attn_backward(hip.BF16, qkv, do, o, lse, dqb, B, T, H, dh, dh ** -0.5); e.record(); torch.cuda.synchronize()
us = a.elapsed_time(e) * 1e3
buf = (ctypes.c_ulonglong * (256 * 16 * 4))()
assert hip.load().nkb_attn_read_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(256, 16, 4).astype(np.int64)
heads_per_cu = B * H / 256.0
print(f"B={B} T={T} H={H}: {us:.1f} us per launch = {us / heads_per_cu:.2f} us per head and CU (the last 256 workgroups are stamped)")
t0 = st[:, :, 0].min(axis=1, keepdims=True)
rel = st[:, :, :] - t0[:, :, None]
tick = float(np.median(rel[:, :, 3].max(axis=1))) / (us / heads_per_cu)     # ticks per us, taking a workgroup's span as the per-head time
for name, k in (("prologue done", 1), ("pass A done", 2), ("pass B done", 3)):
    v = rel[:, :, k] / tick
    print(f"  {name:14s}: median over workgroups of the per-wave times (us): " + " ".join(f"{np.median(v[:, w]):5.2f}" for w in range(16)))
print(f"  ({tick:.0f} ticks per us by that calibration)")
print(f"  workgroup span (last wave's pass B - first stamp): median {np.median(rel[:, :, 3].max(axis=1)) / tick:.2f} us")
