"""Run-to-run identity of wgradr / wgrad3x3 while another stream keeps the GPU busy (the train step's situation): N launches of one shape on a
side stream, every result compared with the first.  `python scripts/wr_race.py [M Cin Cout]`"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
dev = "cuda"; T = torch.bfloat16; d = hip.BF16
shapes = [(50176, 1024, 256), (50176, 256, 1024), (12544, 2048, 512), (200704, 512, 128), (802816, 256, 128)]
if len(sys.argv) > 3: shapes = [tuple(int(a) for a in sys.argv[1:4])]
side = torch.cuda.Stream()
big = torch.randn(64 * 1024 * 1024, device=dev)
mm_a = torch.randn(4096, 4096, device=dev, dtype=T); mm_b = torch.randn(4096, 4096, device=dev, dtype=T)
for M, K, N in shapes:
    x = torch.randn(M, K, device=dev).to(T); dy = torch.randn(M, N, device=dev).to(T)
    ws = torch.empty(hip.conv_wgrad_workspace(d, N=M, P=1, Q=1, Cin=K, Cout=N), device=dev)
    outs = []
    torch.cuda.synchronize()
    for it in range(40):
        dw = torch.zeros(N, K, device=dev)
        torch.cuda.synchronize()
        # main stream: bandwidth + MFMA load of varying length
        for _ in range(it % 4):
            big.mul_(1.0001); (mm_a @ mm_b)
        with torch.cuda.stream(side):
            hip.conv_wgrad(d, dy, x, dw, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N, workspace=ws)
        for _ in range(3):
            big.mul_(0.9999); (mm_a @ mm_b)
        outs.append(dw)
    torch.cuda.synchronize()
    bad = [i for i, o in enumerate(outs) if not torch.equal(o, outs[0])]
    diffs = [(outs[i] - outs[0]).abs().max().item() for i in bad[:3]]
    print(f"M={M} Cin={K} Cout={N}: {len(bad)} of {len(outs)} launches differ from the first {diffs}", flush=True)
