"""Five launches of the ring-buffered stem convolution (for rocprofv3 counter passes): python one_stemp.py <N> <H> <W>"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
N, H, W = [int(v) for v in sys.argv[1:4]]
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
Wp = (W + 1) & ~1; P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
xp = torch.randn(N, H, Wp, 4, device=dev).to(T); wp = (torch.randn(64, hip.stem_weight_cols(d), device=dev) * 0.05).to(T)
y = torch.empty(N, P, Q, 64, device=dev, dtype=T)
tiles = hip.stemp_tiles(d, N, H, W, 64)
st = torch.zeros(hip.bn_stats_floats(tiles, 64), device=dev)
for _ in range(5):
    hip.stemp_conv(d, xp, wp, y, st, N, H, W, 64, 64)
torch.cuda.synchronize()
