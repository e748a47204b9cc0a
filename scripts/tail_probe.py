"""What the remainder tiles of a conv grid cost: the layer3 3x3 (256 -> 256, 14 x 14) and layer2 3x3 (128 -> 128, 28 x 28) forward
launches at batch sizes around the point where the tile count crosses a multiple of the 768 resident workgroups (3 per CU)."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (c, h, batches) in [(256, 14, (244, 248, 250, 252, 256, 260)), (128, 28, (122, 125, 126, 128, 130)), (64, 56, (240, 244, 248, 252, 256))]:
    for B in batches:
        x = torch.randn(B, h, h, c, device=dev).to(T); w = (torch.randn(c, 3, 3, c, device=dev) * 0.05).to(T)
        y = torch.empty(B, h, h, c, device=dev, dtype=T)
        stats = torch.zeros(hip.bn_stats_floats(hip.stat_tiles(d, B * h * h, c), c), device=dev)
        t = timeit(lambda: hip.conv_gemm(d, 0, x, w, y, N=B, H=h, W=h, Cin=c, ldx=c, P=h, Q=h, Cout=c, ldy=c, R=3, S=3, stride=1, pad=1, stats=stats))
        M = B * h * h
        tiles = -(-M // 128) * max(c // 128, 1) if c > 64 else -(-M // 256)
        print(f"3x3 {c:3d} ch {h:2d}x{h:<2d} bs {B:3d}: M {M:6d} tiles {tiles:5d} = {tiles / 768:5.2f} x 768  {t:7.1f} us  {2.0 * M * c * c * 9 / t / 1e6:6.1f} TFLOP/s", flush=True)
