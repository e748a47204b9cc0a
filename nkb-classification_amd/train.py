"""Config-driven training entry point — drop-in for /root/reference/train.py.

    python train.py -cfg configs/my_config.py

Same epoch driver as train.py:19-73: TrainLogger, GradScaler(enabled=cfg.enable_gradient_scaler), per-epoch
backbone freeze policy, train_epoch / val_epoch, compute_metrics, best-by-validation-balanced-accuracy and last
checkpoints under <experiment>/weights/.  Checkpoints are state dicts with timm-compatible keys
(`best.pth`, `last.pth`) plus the TorchScript archives `scripted_best.pt` / `scripted_last.pt` of train.py:66-73,
scripted from a plain-torch module with the same architecture and weights (nkb_classification/scripted.py).
Launch under `torch.distributed.run` to train data-parallel (one process per GPU, RCCL gradient all-reduce): the reference is
single-device (train.py:98), so everything rank-aware here is new — rank 0's parameters and BatchNorm buffers are broadcast
before the first step, every rank reads its own shard of the data (`rank` / `world` injected into cfg.train_data /
cfg.val_data), the per-epoch result lists are gathered from all ranks before `compute_metrics`, and only rank 0 creates the
experiment directory, writes metrics.csv / classes.json and saves checkpoints.  Optional config keys: `seed` (default 0).
Environment (tests / rehearsals): NKB_DDP_BACKEND (default "nccl" = RCCL), NKB_DDP_ONE_GPU=1 (every rank on cuda:0).
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
from nkb_classification.amp import HipGradScaler  # noqa: E402
from nkb_classification import parallel  # noqa: E402


def train(model, train_loader, val_loader, optimizer, scheduler, criterion, comet_experiment, local_experiment, device,
          cfg):
    """Epoch driver with the reference's signature (train.py:19-73).  local_experiment is None on ranks other than 0."""
    rank0 = parallel.rank() == 0
    model_path = local_experiment.path / "weights" if local_experiment is not None else None
    best_val_acc = 0
    classes = train_loader.dataset.classes
    train_logger = TrainLogger(cfg, comet_experiment, local_experiment, classes)
    train_logger.log_images_at_start(train_loader)
    scaler = HipGradScaler("cuda", enabled=cfg.enable_gradient_scaler)
    train_results = val_results = None

    for epoch in tqdm(range(cfg.n_epochs), desc="Training epochs", disable=not rank0):
        if epoch in cfg.backbone_state_policy.keys():
            model.set_backbone_state(cfg.backbone_state_policy[epoch])
        for loader in (train_loader, val_loader):
            sampler = getattr(getattr(loader, "loader", loader), "sampler", None)
            if hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)                 # data parallel: the shared permutation is re-drawn per epoch
        train_results = parallel.gather_epoch_results(
            train_epoch(model, train_loader, optimizer, scheduler, scaler, criterion, device, cfg, train_logger),
            real_len=_real_len(train_loader))
        scaler.settle()          # a skipped LAST step must not stay counted in the optimizer state the checkpoint sees
        val_results = parallel.gather_epoch_results(val_epoch(model, val_loader, criterion, device, cfg, train_logger),
                                                    real_len=_real_len(val_loader))
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
    return train_results, val_results


def _real_len(loader):
    """Samples of this rank's shard that are not padding repeats (None: not a sharded loader)."""
    sampler = getattr(getattr(loader, "loader", loader), "sampler", None)
    return getattr(sampler, "real_len", None)


_after_train = None      # callable(model=, train_loader=, val_loader=, results=, cfg=): set by a wrapper that imports this module


def main():
    parser = argparse.ArgumentParser(description="Train arguments")
    parser.add_argument("-cfg", "--config", help="Config file path", type=str, default="", required=True)
    args = parser.parse_args()
    exec(read_py_config(args.config), globals(), globals())
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    device = torch.device(cfg.device)  # noqa: F821 (cfg is bound by the exec above, as in the reference)
    if world > 1:
        local = 0 if os.environ.get("NKB_DDP_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", 0))
        device = torch.device("cuda", local)
    if device.type == "cuda":
        torch.cuda.set_device(device)          # hip.stream() is the current device's stream: make it the model's device
    if world > 1:
        backend = os.environ.get("NKB_DDP_BACKEND", "nccl")
        torch.distributed.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    seed = int(getattr(cfg, "seed", 0))  # noqa: F821
    torch.manual_seed(seed)                    # same initial weights everywhere even before the broadcast below
    shard = {"rank": rank, "world": world, "shard_seed": seed} if world > 1 else {}
    train_loader = get_dataset({**cfg.train_data, **shard}, getattr(cfg, "train_pipeline", None))  # noqa: F821
    classes = train_loader.dataset.classes
    val_data = cfg.val_data if "classes" in cfg.val_data.keys() else {**cfg.val_data, "classes": classes}  # noqa: F821
    # validation shards stay unpadded: no collective runs inside val_epoch, and every image must count exactly once
    val_loader = get_dataset({**val_data, **shard, **({"shard_pad": False} if world > 1 else {})}, getattr(cfg, "val_pipeline", None))
    model = get_model(cfg.model, classes, device, compile=cfg.compile)  # noqa: F821
    optimizer = get_optimizer(model, cfg_optimizer=cfg.optimizer)  # noqa: F821
    scheduler = get_scheduler(optimizer, cfg.lr_policy)  # noqa: F821
    criterion = get_loss(cfg.criterion, device)  # noqa: F821
    if world > 1:
        parallel.attach(model, optimizer, device)
        torch.manual_seed(seed + 1 + rank)     # dropout / stochastic-depth draws differ per rank from here on
    # train.py:104-108: only rank 0 owns the run directory (no exists()->mkdir race, one metrics.csv)
    local_experiment = get_local_experiment(cfg.experiment["local"]) if rank == 0 else None  # noqa: F821
    out = train(model, train_loader, val_loader, optimizer, scheduler, criterion, None, local_experiment, device, cfg)  # noqa: F821
    if _after_train is not None:
        _after_train(model=model, train_loader=train_loader, val_loader=val_loader, results=out, cfg=cfg)  # noqa: F821
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
