"""Ring-buffered stem convolution (csrc/stemp.hip) next to nkb_stem_conv on the same packed operands: outputs, partial sums, times.
Usage: python scripts/stemp_check.py [quick]"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip  # noqa: E402

DEV = torch.device("cuda", 0)
d = hip.BF16


def run(N, H, W, reps):
    torch.manual_seed(N * 1000 + H + W)
    Co = 64
    img = torch.randn(N, 3, H, W, device=DEV)
    w = torch.randn(Co, 7, 7, 3, device=DEV) / (147 ** 0.5)
    Wp = (W + 1) & ~1
    xp = torch.empty(N, H, Wp, 4, device=DEV, dtype=torch.bfloat16)
    hip.stem_pack(d, img, xp, N, 3, H, W)
    wp = torch.empty(Co, hip.stem_weight_cols(d), device=DEV, dtype=torch.bfloat16)
    hip.stem_wprep(d, w, wp, Co, 3)
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    M = N * P * Q
    tiles0 = hip.stat_tiles(d, M, Co)
    tiles1 = hip.stemp_tiles(d, N, H, W, Co)
    assert tiles1 > 0, "not eligible"
    y0 = torch.empty(N, P, Q, Co, device=DEV, dtype=torch.bfloat16)
    y1 = torch.full_like(y0, float("nan"))
    s0 = torch.zeros(hip.bn_stats_floats(tiles0, Co), device=DEV)
    s1 = torch.full((hip.bn_stats_floats(tiles1, Co),), float("nan"), device=DEV)
    f0 = lambda: hip.stem_conv(d, xp, wp, y0, s0, N, H, W, Co, Co)        # noqa: E731
    f1 = lambda: hip.stemp_conv(d, xp, wp, y1, s1, N, H, W, Co, Co)       # noqa: E731
    f0(); f1()
    torch.cuda.synchronize()
    a, b = y0.float(), y1.float()
    bad = (~torch.isfinite(b)).sum().item()
    err = (a - b).abs().max().item()
    ref = a.abs().max().item()
    t0 = s0[: tiles0 * 2 * Co].view(tiles0, 2, Co).double().sum(0)
    t1 = s1[: tiles1 * 2 * Co].view(tiles1, 2, Co).double().sum(0)
    serr = ((t0 - t1).abs() / (t0.abs() + 1e-3 * t0.abs().max() + 1e-6)).max().item()
    mism = ((a - b).abs() > 0.02 * ref).sum().item()
    times = [[], []]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for r in range(reps):
        for k, f in enumerate((f0, f1)):
            ev[0].record(); f(); ev[1].record(); torch.cuda.synchronize()
            times[k].append(ev[0].elapsed_time(ev[1]) * 1e3)
    med = [sorted(t)[len(t) // 2] if t else 0.0 for t in times]
    mb = (N * H * Wp * 4 + M * Co) * 2 / 1e6
    print(f"N={N:3d} {H:3d}x{W:<3d} tiles {tiles0:5d}/{tiles1:3d}  max|dy| {err:.3e} (ref {ref:.2f}) nonfinite {bad} outliers {mism}  "
          f"stats rel {serr:.2e}  old {med[0]:7.1f} us  new {med[1]:7.1f} us  ({mb / max(med[1], 1e-9):.2f} TB/s)", flush=True)
    ok = bad == 0 and mism == 0 and err <= 0.02 * ref + 1e-3 and serr < 2e-2
    # weight gradient: the ring kernel next to the generic split-over-pixels kernel (different summation orders: fp32 tolerance)
    dy = torch.randn(N, P, Q, Co, device=DEV).to(torch.bfloat16)
    cols = hip.stem_weight_cols(d) - 32
    dw0 = torch.zeros(Co, cols, device=DEV)
    dw1 = torch.zeros(Co, cols, device=DEV)
    work0 = torch.empty(hip.stem_wgrad_workspace(d, N, H, W, Co), device=DEV)
    work1 = torch.empty(hip.stemp_wgrad_workspace(d, N, H, W, Co), device=DEV)
    g0 = lambda: hip.stem_wgrad(d, dy, xp, dw0, N, H, W, Co, Co, workspace=work0)      # noqa: E731
    g1 = lambda: hip.stemp_wgrad(d, dy, xp, dw1, N, H, W, Co, Co, work1)               # noqa: E731
    g0(); g1()
    torch.cuda.synchronize()
    werr = (dw0 - dw1).abs().max().item()
    wref = dw0.abs().max().item()
    times = [[], []]
    for r in range(reps):
        for k, f in enumerate((g0, g1)):
            ev[0].record(); f(); ev[1].record(); torch.cuda.synchronize()
            times[k].append(ev[0].elapsed_time(ev[1]) * 1e3)
    med = [sorted(t)[len(t) // 2] if t else 0.0 for t in times]
    print(f"      weight gradient: max|d| {werr:.3e} (ref {wref:.1f})  old {med[0]:7.1f} us  new {med[1]:7.1f} us", flush=True)
    return ok and werr <= 2e-3 * wref + 1e-3


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    shapes = [(2, 224, 224), (3, 64, 64), (5, 97, 131), (1, 224, 224), (7, 32, 250)]
    if not quick:
        shapes += [(256, 224, 224), (64, 224, 224)]
    allok = True
    for sh in shapes:
        allok &= run(*sh, 0 if quick else 7)
    print("ALL OK" if allok else "MISMATCH")
    sys.exit(0 if allok else 1)
