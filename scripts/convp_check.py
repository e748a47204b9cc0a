"""Row-balanced 3x3 core (csrc/convp.hip) next to the 128 x 128 implicit-GEMM kernel on the same operands: outputs, partial sums,
and the time of both (HIP events, alternating).  Usage: python scripts/convp_check.py [quick]"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip  # noqa: E402

DEV = torch.device("cuda", 0)
d = hip.BF16


def run(N, H, W, Cin, Cout, kind, reps):
    torch.manual_seed(N * 1000 + H + Cin + kind)
    M = N * H * W
    x = (torch.randn(N, H, W, Cin, device=DEV)).to(torch.bfloat16)
    w = (torch.randn(Cout, 3, 3, Cin, device=DEV) * (1.0 / (3 * Cin ** 0.5))).to(torch.bfloat16)
    tiles0 = hip.stat_tiles(d, M, Cout)
    tiles1 = hip.convp_config(True, True, True, True) or hip.convp_tiles(d, kind, N=N, H=H, W=W, Cin=Cin, ldx=Cin, Cout=Cout, ldy=Cout, R=3, S=3, stride=1, pad=1)
    assert tiles1 > 0, "not eligible"
    y0 = torch.empty(N, H, W, Cout, device=DEV, dtype=torch.bfloat16)
    y1 = torch.full_like(y0, float("nan"))
    s0 = torch.zeros(hip.bn_stats_floats(tiles0, Cout), device=DEV)
    s1 = torch.full((hip.bn_stats_floats(tiles1, Cout),), float("nan"), device=DEV)
    geom = dict(N=N, H=H, W=W, Cin=Cin, ldx=Cin, P=H, Q=W, Cout=Cout, ldy=Cout, R=3, S=3, stride=1, pad=1)
    if kind == 0:
        f0 = lambda: hip.conv_gemm(d, 0, x, w, y0, stats=s0, **geom)                                     # noqa: E731
        f1 = lambda: hip.convp_fwd(d, x, w, y1, s1, N=N, H=H, W=W, Cin=Cin, ldx=Cin, Cout=Cout, ldy=Cout, tiles=tiles1)  # noqa: E731
    else:
        c = torch.randn(N, H, W, Cout, device=DEV).to(torch.bfloat16)
        scale = (torch.rand(Cout, device=DEV) + 0.5)
        shift = torch.randn(Cout, device=DEV) * 0.3
        mean = torch.randn(Cout, device=DEV) * 0.1
        f0 = lambda: hip.conv_dgrad_bn(d, x, w, y0, c, scale, shift, mean, s0, **geom)                    # noqa: E731
        f1 = lambda: hip.convp_dgrad_bn(d, x, w, y1, c, scale, shift, mean, s1, N=N, H=H, W=W, Cin=Cin, ldx=Cin, Cout=Cout, ldy=Cout, tiles=tiles1)  # noqa: E731
    f0(); f1()
    torch.cuda.synchronize()
    a, b = y0.float(), y1.float()
    bad = (~torch.isfinite(b)).sum().item()
    err = (a - b).abs().max().item()
    ref = a.abs().max().item()
    t0 = s0[: tiles0 * 2 * Cout].view(tiles0, 2, Cout).double().sum(0)
    t1 = s1[: tiles1 * 2 * Cout].view(tiles1, 2, Cout).double().sum(0)
    serr = ((t0 - t1).abs() / (t0.abs() + 1e-3 * t0.abs().max() + 1e-6)).max().item()
    mism = ((a - b).abs() > 0.02 * ref).sum().item()
    times = [[], []]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for r in range(reps):
        for k, f in enumerate((f0, f1)):
            ev[0].record(); f(); ev[1].record(); torch.cuda.synchronize()
            times[k].append(ev[0].elapsed_time(ev[1]) * 1e3)
    med = [sorted(t)[len(t) // 2] if t else 0.0 for t in times]
    gf = 2.0 * M * Cout * 9 * Cin / 1e9
    print(f"kind {kind} N={N:3d} {H:3d}x{W:<3d} {Cin:4d}->{Cout:<4d} tiles {tiles0:5d}/{tiles1:3d}  max|dy| {err:.3e} (ref {ref:.2f}) nonfinite {bad} "
          f"outliers {mism}  stats rel {serr:.2e}  old {med[0]:7.1f} us  new {med[1]:7.1f} us  ({gf / max(med[1], 1e-9):.2f} PF/s)", flush=True)
    ok = bad == 0 and mism == 0 and err <= 0.02 * ref + 1e-3 and serr < 2e-2
    return ok


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    shapes = [(32, 14, 14, 256, 256), (8, 28, 28, 128, 128), (96, 7, 7, 512, 512), (9, 23, 23, 64, 128), (24, 14, 14, 128, 384)]
    shapes += [(4, 56, 56, 64, 64), (5, 37, 29, 64, 64), (2, 48, 48, 64, 64)]          # the resident-filter form (convp64_kernel)
    if not quick:
        shapes += [(256, 28, 28, 128, 128), (256, 14, 14, 256, 256), (256, 7, 7, 512, 512), (256, 56, 56, 64, 64)]
    if len(sys.argv) > 1 and sys.argv[1] == "c64":
        shapes = [(4, 56, 56, 64, 64), (5, 37, 29, 64, 64), (2, 48, 48, 64, 64), (256, 56, 56, 64, 64)]
    allok = True
    for sh in shapes:
        for kind in (0, 1):
            allok &= run(*sh, kind, 0 if quick else 7)
    print("ALL OK" if allok else "MISMATCH")
    sys.exit(0 if allok else 1)
