"""Config-driven training entry point — drop-in for /root/reference/train.py.

    python train.py -cfg configs/my_config.py

Same epoch driver as train.py:19-73: TrainLogger, GradScaler(enabled=cfg.enable_gradient_scaler), per-epoch
backbone freeze policy, train_epoch / val_epoch, compute_metrics, best-by-validation-balanced-accuracy and last
checkpoints under <experiment>/weights/.  Checkpoints are state dicts with timm-compatible keys
(`best.pth`, `last.pth`) plus the TorchScript archives `scripted_best.pt` / `scripted_last.pt` of train.py:66-73,
scripted from a plain-torch module with the same architecture and weights (nkb_classification/scripted.py).
Launch under `torch.distributed.run` to train data-parallel (one process per GPU, RCCL gradient all-reduce).
"""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")            # before the HIP runtime starts: see nkb_classification/__init__.py
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
from tqdm import tqdm  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent))

from nkb_classification.dataset import get_dataset  # noqa: E402
from nkb_classification.engine import train_epoch, val_epoch  # noqa: E402
from nkb_classification.logging import TrainLogger, get_local_experiment  # noqa: E402
from nkb_classification.losses import get_loss  # noqa: E402
from nkb_classification.metrics import compute_metrics  # noqa: E402
from nkb_classification.model import get_model  # noqa: E402
from nkb_classification.scripted import save_scripted  # noqa: E402
from nkb_classification.utils import get_optimizer, get_scheduler, read_py_config  # noqa: E402


def train(model, train_loader, val_loader, optimizer, scheduler, criterion, comet_experiment, local_experiment, device,
          cfg):
    model_path = local_experiment.path / "weights"
    best_val_acc = 0
    classes = train_loader.dataset.classes
    train_logger = TrainLogger(cfg, comet_experiment, local_experiment, classes)
    train_logger.log_images_at_start(train_loader)
    scaler = torch.amp.GradScaler("cuda", enabled=cfg.enable_gradient_scaler)
    rank0 = (not torch.distributed.is_initialized()) or torch.distributed.get_rank() == 0

    for epoch in tqdm(range(cfg.n_epochs), desc="Training epochs"):
        if epoch in cfg.backbone_state_policy.keys():
            model.set_backbone_state(cfg.backbone_state_policy[epoch])
        train_results = train_epoch(model, train_loader, optimizer, scheduler, scaler, criterion, device, cfg, train_logger)
        val_results = val_epoch(model, val_loader, criterion, device, cfg, train_logger)
        train_results["metrics"] = compute_metrics(cfg, train_results)
        val_results["metrics"] = compute_metrics(cfg, val_results)
        epoch_val_acc = val_results["metrics"]["epoch_acc"]
        train_logger.log_epoch(epoch, train_results, val_results)
        if not rank0:
            continue
        if epoch_val_acc is not None and epoch_val_acc > best_val_acc:
            best_val_acc = epoch_val_acc
            torch.save(model.state_dict(), Path(model_path, "best.pth"))
            save_scripted(model, Path(model_path, "scripted_best.pt"))
        torch.save(model.state_dict(), Path(model_path, "last.pth"))
        save_scripted(model, Path(model_path, "scripted_last.pt"))


def main():
    parser = argparse.ArgumentParser(description="Train arguments")
    parser.add_argument("-cfg", "--config", help="Config file path", type=str, default="", required=True)
    args = parser.parse_args()
    exec(read_py_config(args.config), globals(), globals())
    world = int(os.environ.get("WORLD_SIZE", 1))
    device = torch.device(cfg.device)  # noqa: F821 (cfg is bound by the exec above, as in the reference)
    if world > 1:
        local = int(os.environ.get("LOCAL_RANK", 0))
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
        torch.distributed.init_process_group("nccl", device_id=device)
    train_loader = get_dataset(cfg.train_data, getattr(cfg, "train_pipeline", None))  # noqa: F821
    classes = train_loader.dataset.classes
    if "classes" not in cfg.val_data.keys():  # noqa: F821
        cfg.val_data = {**cfg.val_data, "classes": classes}  # noqa: F821
    val_loader = get_dataset(cfg.val_data, getattr(cfg, "val_pipeline", None))  # noqa: F821
    model = get_model(cfg.model, classes, device, compile=cfg.compile)  # noqa: F821
    optimizer = get_optimizer(model, cfg_optimizer=cfg.optimizer)  # noqa: F821
    scheduler = get_scheduler(optimizer, cfg.lr_policy)  # noqa: F821
    criterion = get_loss(cfg.criterion, device)  # noqa: F821
    if world > 1:
        from nkb_classification.parallel import GradReducer
        GradReducer(model, optimizer)
    local_experiment = get_local_experiment(cfg.experiment["local"])  # noqa: F821
    train(model, train_loader, val_loader, optimizer, scheduler, criterion, None, local_experiment, device, cfg)  # noqa: F821
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
