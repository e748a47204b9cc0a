"""Streaming R = g^T a kernel (csrc/gramr.hip) next to the generic weight-gradient kernel (nkb_conv_wgrad_assign) on the same operands.
Usage: python scripts/gramr_check.py [quick]"""
import os
import sys

import torch

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
sys.path.insert(0, os.path.join(R_, "nkb-classification_amd"))
from nkb_classification import hip  # noqa: E402

DEV = torch.device("cuda", 0)
d = hip.BF16


def run(N, H, co, ci, reps):
    torch.manual_seed(N + H + co)
    M = N * H * H
    g = torch.randn(M, co, device=DEV).to(torch.bfloat16)
    a = torch.randn(M, ci, device=DEV).to(torch.bfloat16)
    need = hip.gramr_workspace(d, M, co, ci)
    assert need > 0, "not eligible"
    R0 = torch.full((co, ci), float("nan"), device=DEV)
    R1 = torch.full((co, ci), float("nan"), device=DEV)
    w0 = torch.empty(hip.conv_wgrad_workspace(d, N=N, P=H, Q=H, Cin=ci, Cout=co), device=DEV)
    w1 = torch.empty(need, device=DEV)
    f0 = lambda: hip.conv_wgrad(d, g, a, R0, N=N, H=H, W=H, Cin=ci, ldx=ci, P=H, Q=H, Cout=co, lddy=co, workspace=w0, assign=True)   # noqa: E731
    f1 = lambda: hip.gramr(d, g, co, a, ci, R1, M, co, ci, w1)                                                                      # noqa: E731
    f0(); f1()
    torch.cuda.synchronize()
    ref = (g[: min(M, 1 << 16)].float().t() @ a[: min(M, 1 << 16)].float()) if M <= (1 << 16) else None
    err = (R0 - R1).abs().max().item()
    scale = R0.abs().max().item()
    bad = (~torch.isfinite(R1)).sum().item()
    e2 = (ref - R1).abs().max().item() if ref is not None else float("nan")
    times = [[], []]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for r in range(reps):
        for k, f in enumerate((f0, f1)):
            ev[0].record(); f(); ev[1].record(); torch.cuda.synchronize()
            times[k].append(ev[0].elapsed_time(ev[1]) * 1e3)
    med = [sorted(t)[len(t) // 2] if t else 0.0 for t in times]
    mb = M * (co + ci) * 2 / 1e6
    print(f"M={M:7d} {co:4d}x{ci:<4d} max|d| vs generic {err:.3e} vs torch {e2:.3e} (scale {scale:.1f}) nonfinite {bad}  old {med[0]:7.1f} us  "
          f"new {med[1]:7.1f} us  ({mb / max(med[1], 1e-9):.2f} TB/s)", flush=True)
    ok = bad == 0 and err <= 1e-4 * scale + 1e-3
    # the transposed, accumulating form: dW[ci][co] += a^T g (a 1x1 convolution co -> ci with g as its input)
    D0 = torch.ones(ci, co, device=DEV)
    D1 = torch.ones(ci, co, device=DEV)
    hip.conv_wgrad(d, a, g, D0, N=N, H=H, W=H, Cin=co, ldx=co, P=H, Q=H, Cout=ci, lddy=ci,
                   workspace=torch.empty(hip.conv_wgrad_workspace(d, N=N, P=H, Q=H, Cin=co, Cout=ci), device=DEV))
    hip.gramr(d, g, co, a, ci, D1, M, co, ci, w1, assign=False, transposed=True)
    torch.cuda.synchronize()
    terr = (D0 - D1).abs().max().item()
    print(f"          transposed + accumulate: max|d| {terr:.3e}", flush=True)
    return ok and terr <= 1e-4 * scale + 1e-3


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    shapes = [(8, 56, 256, 64), (7, 53, 256, 64), (32, 28, 512, 128), (33, 27, 512, 128)]
    if not quick:
        shapes += [(256, 56, 256, 64), (256, 28, 512, 128)]
    allok = True
    for sh in shapes:
        allok &= run(*sh, 0 if quick else 7)
    print("ALL OK" if allok else "MISMATCH")
    sys.exit(0 if allok else 1)
