"""Config-driven evaluation entry point — drop-in for /root/reference/eval.py (-cfg <config.py>).

    python eval.py -cfg configs/my_eval_config.py

What the reference's script does (eval.py:16-52), on this engine: the validation loader of `cfg.val_data` / `cfg.val_pipeline`
(which must name its classes unless it is an ImageFolder), `get_model(cfg.model, classes, device)` — normally with
`cfg.model["checkpoint"]` pointing at a `best.pth` / `last.pth` (or a scripted archive) written by train.py — one `val_epoch`
under a `BaseLogger`, `compute_metrics`, and `<cfg.save_path>/metrics.json`.  In eval mode the ResNet family runs the folded
one-launch-per-stage forward (BatchNorm folded into the filters once per eval phase, nothing saved); the transformer families
run their training kernels without saving activations.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")            # before the HIP runtime starts: see nkb_classification/__init__.py

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent))

from nkb_classification.dataset import get_dataset  # noqa: E402
from nkb_classification.engine import val_epoch  # noqa: E402
from nkb_classification.logging import BaseLogger  # noqa: E402
from nkb_classification.losses import get_loss  # noqa: E402
from nkb_classification.metrics import compute_metrics  # noqa: E402
from nkb_classification.model import get_model  # noqa: E402
from nkb_classification.utils import read_py_config  # noqa: E402


def to_plain(obj):
    """numpy / torch scalars and arrays -> JSON types, recursively (what metrics.json may hold)."""
    if isinstance(obj, dict):
        return {str(k): to_plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_plain(v) for v in obj]
    if isinstance(obj, torch.Tensor):
        return to_plain(obj.detach().cpu().numpy())
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def evaluate(model, val_loader, criterion, device, cfg):
    """One validation epoch and its metrics (eval.py:16-24)."""
    logger = BaseLogger(cfg, val_loader.dataset.classes)
    return compute_metrics(cfg, val_epoch(model, val_loader, criterion, device, cfg, logger))


def main():
    parser = argparse.ArgumentParser(description="Evaluation arguments")
    parser.add_argument("-cfg", "--config", help="Config file path", type=str, default="", required=True)
    args = parser.parse_args()
    exec(read_py_config(args.config), globals(), globals())
    if not ("classes" in cfg.val_data or cfg.val_data["type"] == "ImageFolder"):  # noqa: F821 (cfg is bound by the exec above)
        raise SystemExit("eval.py: cfg.val_data must list its classes (or be an ImageFolder)")
    device = torch.device(cfg.device)  # noqa: F821
    if device.type == "cuda":
        torch.cuda.set_device(device)
    val_loader = get_dataset(cfg.val_data, getattr(cfg, "val_pipeline", None))  # noqa: F821
    classes = val_loader.dataset.classes
    model = get_model(cfg.model, classes, device, compile=getattr(cfg, "compile", False))  # noqa: F821
    criterion = get_loss(cfg.criterion, device)  # noqa: F821
    metrics = evaluate(model, val_loader, criterion, device, cfg)  # noqa: F821
    save_path = Path(cfg.save_path)  # noqa: F821
    save_path.mkdir(exist_ok=True, parents=True)
    with open(save_path / "metrics.json", "w") as f:
        json.dump(to_plain(metrics), f)


if __name__ == "__main__":
    main()
