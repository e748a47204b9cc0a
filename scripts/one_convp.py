"""Five launches of the row-balanced 3x3 core on one shape (for rocprofv3 counter passes): python one_convp.py <N> <H> <Cin> <Cout> <kind>"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
N, H, ci, co, kind = [int(v) for v in sys.argv[1:6]]
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
x = torch.randn(N, H, H, ci, device=dev).to(T); w = (torch.randn(co, 3, 3, ci, device=dev) * 0.05).to(T)
y = torch.empty(N, H, H, co, device=dev, dtype=T)
tiles = hip.convp_tiles(d, kind, N=N, H=H, W=H, Cin=ci, ldx=ci, Cout=co, ldy=co, R=3, S=3, stride=1, pad=1)
st = torch.zeros(hip.bn_stats_floats(tiles, co), device=dev)
c = torch.randn(N, H, H, co, device=dev).to(T); sc = torch.ones(co, device=dev); sh = torch.zeros(co, device=dev)
for _ in range(5):
    if kind == 0: hip.convp_fwd(d, x, w, y, st, N=N, H=H, W=H, Cin=ci, ldx=ci, Cout=co, ldy=co, tiles=tiles)
    else: hip.convp_dgrad_bn(d, x, w, y, c, sc, sh, sh, st, N=N, H=H, W=H, Cin=ci, ldx=ci, Cout=co, ldy=co, tiles=tiles)
torch.cuda.synchronize()
