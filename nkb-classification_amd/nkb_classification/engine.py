"""Train / validation epoch drivers — drop-in for /root/reference/nkb_classification/engine.py:20-117.

Same positional signatures and the same step order (zero_grad -> forward -> loss -> backward -> optimizer
step -> log; scheduler stepped once per epoch), the same returned dict.  Differences are confined to where the
host waits for the device:
  * the image batch is uploaded with a non-blocking copy,
  * the progress bar prints the PREVIOUS step's loss, so `loss.item()` (engine.py:13-17, issued between
    forward and backward in the reference) never stalls the step that is being enqueued,
  * the logger keeps its per-step tensors on the device (logging.py in this package).
Mixed precision: `cfg.enable_mixed_presicion` selects bf16 compute (fp32 accumulate / statistics / master
weights) through torch.autocast, the mechanism the reference uses for its fp16 mode (engine.py:43-47).
"""
from __future__ import annotations

from collections import defaultdict

import torch
from tqdm import tqdm

from . import hip


class TrainPbar(tqdm):
    def __init__(self, train_loader, leave, desc, cfg):
        super().__init__(train_loader, leave=leave, desc=desc)
        self.cfg = cfg
        self._pending = None

    def update_loss(self, loss):
        prev, self._pending = self._pending, loss
        if prev is None:
            return
        if self.cfg.task == "multi" and self.cfg.show_full_current_loss_in_terminal:
            self.set_postfix_str(", ".join(f"loss {k}: {v:.4f}" for k, v in prev.items()), refresh=False)
        elif self.cfg.task == "multi":
            self.set_postfix_str(f"Loss: {prev['loss'].item():.4f}", refresh=False)
        else:
            self.set_postfix_str(f"Loss: {prev.item():.4f}", refresh=False)


def _amp_dtype(cfg):
    """cfg.amp_dtype: torch.bfloat16 (default) — or "fp8": bf16 autocast with the Linear contractions of transformer blocks
    in per-tensor-scaled fp8 (BASELINE configs[4]; the reference itself only has fp16 autocast, engine.py:43-47)."""
    d = getattr(cfg, "amp_dtype", torch.bfloat16)
    return torch.bfloat16 if d == "fp8" else d


def _grad_norms(model, log):
    """engine.py:64-73: per-parameter gradient L2 norms + their sum, computed by one segmented HIP reduction
    over the flat gradient arena when the model has one."""
    arena = getattr(model, "arena", None)
    named = [(tag, p) for tag, p in model.named_parameters() if p.grad is not None]
    for tag, _ in named:
        assert tag != "Total"
    if arena is not None and arena.packed and all(arena.owns(p) for _, p in named) and named:
        offs = []
        for _, p in named:
            o = arena.offset_of(p)
            offs += [o, o + p.numel()]
        dev = arena.flat_grad.device
        # segments are [o_i, o_i + n_i); encoded as consecutive pairs -> 2k-1 segments, odd ones are the gaps
        offsets = torch.tensor(offs, dtype=torch.int64, device=dev)
        out = torch.empty(len(offs) - 1, device=dev, dtype=torch.float32)
        hip.segment_sumsq(arena.flat_grad, offsets, len(offs) - 1, out)
        norms = out[0::2].sqrt()
    else:
        norms = torch.stack([p.grad.norm() for _, p in named]) if named else torch.zeros(0)
    total = norms.sum() if named else 0
    for i, (tag, _) in enumerate(named):
        log[f"Gradients/{tag}"].append(norms[i])
    log["Gradients/Total"].append(total)


def _upload(target, device):
    """Labels follow the image to the device (engine.py:41 leaves dict targets to the criterion / logger)."""
    if isinstance(target, dict):
        return {name: labels.to(device, non_blocking=True) for name, labels in target.items()}
    return target.to(device, non_blocking=True)


def _forward_and_loss(model, criterion, img, target, device, cfg):
    """engine.py:43-51: autocast region around model(img) and criterion(preds, target)."""
    if hasattr(model, "fp8_linear"):
        model.fp8_linear = bool(cfg.enable_mixed_presicion and getattr(cfg, "amp_dtype", None) == "fp8")
    with torch.autocast(device_type="cuda", dtype=_amp_dtype(cfg), enabled=cfg.enable_mixed_presicion):
        preds = model(img)
        labels = _upload(target, device) if isinstance(target, torch.Tensor) else target
        return preds, labels, criterion(preds, labels)


def train_epoch(model, train_loader, optimizer, scheduler, scaler, criterion, device, cfg, epoch_logger):
    """One training epoch (engine.py:20-85), positional arguments in the reference's order."""
    model.train()
    epoch_logger.init_iter_logs()
    grad_log = defaultdict(list) if cfg.log_gradients else None
    bar = TrainPbar(train_loader, leave=False, desc="Training", cfg=cfg)
    for img, target in bar:
        img = img.to(device, non_blocking=True)
        optimizer.zero_grad()
        preds, target, loss = _forward_and_loss(model, criterion, img, target, device, cfg)
        bar.update_loss(loss)                              # prints the previous step's loss: no sync between fwd and bwd
        total = loss["loss"] if isinstance(loss, dict) else loss
        scaler.scale(total).backward()
        scaler.step(optimizer)
        scaler.update()
        epoch_logger.log_iter(preds, _upload(target, device) if isinstance(target, dict) else target, loss)
        if grad_log is not None:
            _grad_norms(model, grad_log)
        epoch_logger.log_images_if_needed(img)
    if scheduler is not None:                              # per-epoch policy (engine.py:77-78)
        scheduler.step()
    results = epoch_logger.get_epoch_results()
    if grad_log is not None:
        results["metrics_grad_log"] = grad_log
    return results


@torch.no_grad()
def val_epoch(model, val_loader, criterion, device, cfg, epoch_logger):
    """One evaluation pass (engine.py:88-117): forward + loss in eval mode, same logger protocol."""
    model.eval()
    epoch_logger.init_iter_logs()
    for img, target in tqdm(val_loader, leave=False, desc="Evaluating"):
        img = img.to(device, non_blocking=True)
        preds, target, loss = _forward_and_loss(model, criterion, img, target, device, cfg)
        epoch_logger.log_iter(preds, _upload(target, device) if isinstance(target, dict) else target, loss)
        epoch_logger.log_images_if_needed(img)
    return epoch_logger.get_epoch_results()
