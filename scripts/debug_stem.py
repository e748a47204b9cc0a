"""Packed stem vs torch on a few shapes (debug helper)."""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nkb-classification_amd"))
from nkb_classification import hip  # noqa: E402

DEV = "cuda:0"
for dtype in (torch.float32, torch.bfloat16):
    for (N, H, W) in [(3, 32, 32), (4, 64, 64), (2, 96, 96), (4, 64, 32), (4, 32, 64)]:
        torch.manual_seed(0)
        d = hip.dt(dtype)
        C, Co = 3, 64
        x = torch.randn(N, C, H, W).to(dtype).float()
        w = (torch.randn(Co, C, 7, 7) * 0.1).to(dtype).float()
        y = F.conv2d(x, w, None, 2, 3)
        P, Q = y.shape[2:]
        xp = torch.empty(N, H, (W + 1) // 2 * 2, 4, device=DEV, dtype=dtype)
        hip.stem_pack(d, x.to(DEV), xp, N, C, H, W)
        wp = torch.empty(Co, hip.stem_weight_cols(d), device=DEV, dtype=dtype)
        hip.stem_wprep(d, w.permute(0, 2, 3, 1).contiguous().to(DEV), wp, Co, C)
        yd = torch.empty(N, P, Q, Co, device=DEV, dtype=dtype)
        hip.stem_conv(d, xp, wp, yd, None, N, H, W, Co, Co)
        torch.cuda.synchronize()
        err = (yd.float().cpu() - y.permute(0, 2, 3, 1)).abs()
        bad = (err > 0.05).nonzero()
        print(dtype, (N, H, W), "max err", err.max().item(), "bad", len(bad), bad[:4].tolist())
