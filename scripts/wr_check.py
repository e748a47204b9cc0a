"""wgradr (csrc/wgradr.hip) through nkb_conv_wgrad: parity with the fp32 product on both orientations, ragged pixel counts and the atomic
form, a checksum per case, and us per launch on the ResNet-50 bs-256 1x1 shapes.  `python scripts/wr_check.py [time]`"""
import os, sys, hashlib, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
CASES = [(4096, 256, 128, 0), (4096, 128, 256, 0), (5000, 256, 256, 0), (8192 + 33, 512, 128, 8), (4100, 128, 512, 16), (12544, 2048, 512, 0),
         (12544, 512, 2048, 0), (50176, 1024, 256, 0), (50176, 256, 1024, 0), (4096 + 31, 384, 256, 0), (6000, 256, 384, 0),
         (70000 + 17, 512, 1024, 0), (70000, 1024, 512, 8), (4096, 256, 128, 0, True), (9000, 384, 768, 8, True), (70000 + 17, 512, 1024, 0, True),
         (50432, 768, 2304, 0, True)]
def run(M, Ci, Co, padc, bias=False, ws=True):
    g = torch.Generator(device="cpu").manual_seed(M + Ci)
    x = torch.randn(M, Ci + padc, generator=g).to(dev, T); dy = torch.randn(M, Co + padc, generator=g).to(dev, T)
    dw = torch.ones(Co, Ci, device=dev)
    db = torch.ones(Co, device=dev) if bias else None
    work = torch.full((hip.conv_wgrad_workspace(d, N=M, P=1, Q=1, Cin=Ci, Cout=Co, has_bias=bias) + 5,), float("nan"), device=dev) if ws else None
    n0 = hip.kernel_launches("wgradr")
    hip.conv_wgrad(d, dy, x, dw, N=M, H=1, W=1, Cin=Ci, ldx=Ci + padc, P=1, Q=1, Cout=Co, lddy=Co + padc, workspace=work, dbias=db)
    torch.cuda.synchronize()
    took = hip.kernel_launches("wgradr") > n0
    ref = dy[:, :Co].float().t() @ x[:, :Ci].float()
    err = (dw - 1.0 - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
    if bias:
        bref = dy[:, :Co].double().sum(0)
        err = max(err, (db.double() - 1.0 - bref).abs().max().item() / (bref.abs().max().item() + 1e-9))
    tail = bool(torch.isnan(work[-5:]).all()) if ws else True
    return dw, err, took, tail
if len(sys.argv) < 2 or sys.argv[1] != "time":
    bad = 0
    for c in CASES:
        dw, err, took, tail = run(*c)
        h = hashlib.md5(dw.cpu().numpy().tobytes()).hexdigest()[:12]
        ok = err < 3e-5 and bool(torch.isfinite(dw).all()) and tail
        bad += not ok
        print(f"{c} wgradr={took} rel err {err:.2e} md5 {h} {'ok' if ok else 'BAD'}", flush=True)
    for c in [(4096, 256, 128, 0), (5000, 128, 256, 0), (4500, 256, 256, 0, True)]:
        dw, err, took, _ = run(*c, ws=False)
        print(f"atomics {c}: wgradr={took} rel err {err:.2e} {'ok' if err < 3e-5 else 'BAD'}")
        bad += err >= 3e-5
    print("FAILED" if bad else "all ok")
    sys.exit(1 if bad else 0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
SHAPES = [("l2 conv1 b0", 802816, 256, 128), ("l2 conv1", 200704, 512, 128), ("l2 conv3", 200704, 128, 512), ("l3 conv1 b0", 200704, 512, 256),
          ("l3 conv1", 50176, 1024, 256), ("l3 conv3", 50176, 256, 1024), ("l4 conv1 b0", 50176, 1024, 512), ("l4 conv1", 12544, 2048, 512),
          ("l4 conv3", 12544, 512, 2048), ("vit qkv", 50432, 768, 2304), ("vit proj", 50432, 768, 768), ("vit fc1", 50432, 768, 3072),
          ("vit fc2", 50432, 3072, 768), ("vitl qkv", 32768, 1024, 3072), ("vitl fc1", 32768, 1024, 4096)]
if len(sys.argv) > 2: SHAPES = [s for s in SHAPES if any(k in s[0] for k in sys.argv[2:])]
for name, M, K, N in SHAPES:
    x = torch.randn(M, K, device=dev).to(T); dy = torch.randn(M, N, device=dev).to(T)
    dw = torch.zeros(N, K, device=dev)
    ws = torch.empty(hip.conv_wgrad_workspace(d, N=M, P=1, Q=1, Cin=K, Cout=N), device=dev)
    n0 = (hip.kernel_launches("wgrad8p"), hip.kernel_launches("wgradr"))
    t = timeit(lambda: hip.conv_wgrad(d, dy, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N, workspace=ws))
    kern = "wgradr" if hip.kernel_launches("wgradr") > n0[1] else "wgrad8p" if hip.kernel_launches("wgrad8p") > n0[0] else "generic"
    print(f"{name:12s} M={M:6d} Cin={K:5d} Cout={N:5d} {kern:8s} {t:7.1f} us | hbm floor {2.0 * M * (K + N) / 5.5e6:6.1f} us  mfma floor "
          f"{2.0 * M * K * N / 1.2e9:6.1f} us | slabs {ws.numel() * 4 / 1e6:6.1f} MB", flush=True)
