"""wgrad3x3 (csrc/wgrad3x3.hip) through nkb_conv_wgrad: parity with torch's fp32 weight gradient on strips of every slot width, ragged
batches and one-k-step launches, a checksum per case (compare across NKB_WGRAD3X3 = 1 / 2 / 3: the kernels are bit-identical), and
us per launch on the four ResNet-50 bs-256 shapes.  `python scripts/w3_check.py [time]`"""
import os, sys, hashlib, torch
import torch.nn.functional as F
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
CASES = [(2, 14, 14, 64, 64), (3, 7, 7, 128, 64), (2, 28, 28, 64, 128), (1, 56, 56, 64, 64), (3, 9, 9, 64, 192), (1, 30, 30, 64, 64),
         (5, 15, 15, 64, 64), (2, 31, 31, 64, 64), (1, 62, 62, 64, 64), (1, 7, 7, 64, 64), (2, 8, 20, 64, 64), (7, 12, 5, 64, 128),
         (33, 14, 14, 128, 128), (16, 7, 7, 192, 64), (9, 28, 28, 64, 64)]
def run(N, H, W, Ci, Co, ws=True):
    g = torch.Generator(device="cpu").manual_seed(N * 1000 + H * 10 + W)
    x = torch.randn(N, H, W, Ci, generator=g).to(dev, T); dy = torch.randn(N, H, W, Co, generator=g).to(dev, T)
    dw = torch.zeros(Co, 3, 3, Ci, device=dev)
    work = torch.empty(hip.conv_wgrad_workspace(d, N=N, P=H, Q=W, Cin=Ci, Cout=Co, R=3, S=3, stride=1, pad=1), device=dev) if ws else None
    n0 = hip.kernel_launches("wgrad3x3")
    hip.conv_wgrad(d, dy, x, dw, N=N, H=H, W=W, Cin=Ci, ldx=Ci, P=H, Q=W, Cout=Co, lddy=Co, R=3, S=3, stride=1, pad=1, workspace=work)
    torch.cuda.synchronize()
    took = hip.kernel_launches("wgrad3x3") > n0
    xf = x.float().permute(0, 3, 1, 2).requires_grad_(False); dyf = dy.float().permute(0, 3, 1, 2)
    wref = torch.nn.grad.conv2d_weight(xf, (Co, Ci, 3, 3), dyf, stride=1, padding=1).permute(0, 2, 3, 1)
    err = (dw - wref).abs().max().item() / (wref.abs().max().item() + 1e-9)
    return dw, err, took
if len(sys.argv) < 2 or sys.argv[1] != "time":
    bad = 0
    for c in CASES:
        dw, err, took = run(*c)
        h = hashlib.md5(dw.cpu().numpy().tobytes()).hexdigest()[:12]
        ok = err < 2e-5 and bool(torch.isfinite(dw).all())
        bad += not ok
        print(f"{c} wgrad3x3={took} rel err {err:.2e} md5 {h} {'ok' if ok else 'BAD'}", flush=True)
    dw, err, took = run(2, 14, 14, 64, 64, ws=False)            # atomic form
    print(f"atomics: rel err {err:.2e} {'ok' if err < 2e-5 else 'BAD'}")
    print("FAILED" if bad or err >= 2e-5 else "all ok")
    sys.exit(1 if bad or err >= 2e-5 else 0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for N, H, W, Ci, Co in [(256, 56, 56, 64, 64), (256, 28, 28, 128, 128), (256, 14, 14, 256, 256), (256, 7, 7, 512, 512)]:
    x = torch.randn(N, H, W, Ci, device=dev).to(T); dy = torch.randn(N, H, W, Co, device=dev).to(T)
    dw = torch.zeros(Co, 3, 3, Ci, device=dev)
    work = torch.empty(hip.conv_wgrad_workspace(d, N=N, P=H, Q=W, Cin=Ci, Cout=Co, R=3, S=3, stride=1, pad=1), device=dev)
    t = timeit(lambda: hip.conv_wgrad(d, dy, x, dw, N=N, H=H, W=W, Cin=Ci, ldx=Ci, P=H, Q=W, Cout=Co, lddy=Co, R=3, S=3, stride=1, pad=1,
                                      workspace=work))
    fl = 2.0 * N * H * W * 9 * Ci * Co
    print(f"{H}x{W} {Ci}->{Co}: {t:7.1f} us (kernel + reduce)  {fl / t / 1e9:6.3f} PFLOP/s  slabs {work.numel() * 4 / 1e6:.1f} MB", flush=True)
