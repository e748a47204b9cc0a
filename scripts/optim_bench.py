"""Fused optimizer step standalone: ms and TB/s (30 bytes per parameter) for ResNet-50 / ViT-B / ViT-L sized flat ranges."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "nkb-classification_amd"))
from nkb_classification import hip
from nkb_classification.utils import _step_scalars
dev = "cuda"
for n in (25_557_032, 86_567_656, 304_000_000):
    p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
    state = {}
    for kind in ("nadam", "adam", "sgd"):
        k, sc = _step_scalars(kind, state, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, momentum_decay=4e-3)
        fn = lambda: hip.optim_step(k, p, g, m, v, sh, n, 1e-3, 0.01, 0.9, 0.999, 1e-8, 1.0, *sc)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        byt = n * (30 if kind != "sgd" else 14)
        print(f"n={n / 1e6:6.1f}M {kind:6s}: {ms * 1e3:8.1f} us  {byt / ms / 1e9:5.2f} TB/s")
    del p, g, m, v, sh
