"""ORACLE (test infrastructure, not product code).

CPU restatement, on stock torch ops, of the reference's step semantics:

* losses        — /root/reference/nkb_classification/losses.py:10-176
* optimizers    — /root/reference/nkb_classification/utils.py:10-61
* logger lists  — /root/reference/nkb_classification/logging.py:245-294
* train / val   — /root/reference/nkb_classification/engine.py:20-117

Pinned against golden vectors captured by importing those reference modules
(oracle/make_golden.py -> tests/golden/*.json), see tests/test_oracle_golden.py.
On CPU the reference's autocast("cuda") and GradScaler disable themselves, so
the numeric mode restated here is plain fp32, unscaled.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, Optional

import torch
from torch import nn
import torch.nn.functional as F


# ---------------------------------------------------------------- losses ----
def cross_entropy(logits: torch.Tensor, y: torch.Tensor, weight: Optional[torch.Tensor] = None):
    """losses.py:158 -> nn.CrossEntropyLoss(weight): weighted mean of -log p[y]."""
    logp = torch.log_softmax(logits.float(), dim=-1)
    nll = -logp.gather(1, y.view(-1, 1)).squeeze(1)
    if weight is None:
        return nll.mean()
    w = weight[y]
    return (nll * w).sum() / w.sum()


def focal_loss(logits, y, alpha=None, gamma=2.0, ignore_index=-100, reduction="mean"):
    """losses.py:59-94."""
    keep = y != ignore_index
    y = y[keep]
    if y.numel() == 0:
        return torch.tensor(0.0)
    logits = logits[keep]
    logp = torch.log_softmax(logits, dim=-1)
    logpt = logp.gather(1, y.view(-1, 1)).squeeze(1)
    ce = -logpt if alpha is None else -logpt * alpha[y]
    loss = (1 - logpt.exp()) ** gamma * ce
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


class Criterion:
    """get_loss (losses.py:154-176) + MultitaskCriterion (losses.py:97-151)."""

    def __init__(self, cfg_loss: dict):
        self.kind = cfg_loss["type"]
        if self.kind not in ("CrossEntropyLoss", "FocalLoss"):
            raise NotImplementedError(f"Unknown loss type in config: {self.kind}")
        self.multi = cfg_loss["task"] == "multi"
        self.weight = torch.tensor(cfg_loss["weight"], dtype=torch.float) if "weight" in cfg_loss else None
        self.alpha = torch.tensor(cfg_loss["alpha"], dtype=torch.float) if "alpha" in cfg_loss else None
        self.gamma = cfg_loss.get("gamma", 2.0)

    def _one(self, x, y):
        if self.kind == "CrossEntropyLoss":
            return cross_entropy(x, y, self.weight)
        return focal_loss(x, y, self.alpha, self.gamma)

    def __call__(self, pred, true):
        if not self.multi:
            return self._one(pred, true)
        assert pred.keys() == true.keys()
        out = {}
        total = 0
        for t in pred:
            out[t] = self._one(pred[t], true[t])
            total = total + out[t]
        out["loss"] = total
        return out


# ------------------------------------------------------------ optimizers ----
def make_optimizer(model, cfg_opt: dict):
    """utils.py:10-42: two parameter groups, per-group lr / weight_decay."""
    lr = cfg_opt.get("lr", 1e-3)
    wd = cfg_opt.get("weight_decay", 0.0)
    groups = [
        dict(params=list(model.emb_model.parameters()), lr=cfg_opt.get("backbone_lr", lr),
             weight_decay=cfg_opt.get("backbone_weight_decay", wd)),
        dict(params=list(model.classifier.parameters()), lr=cfg_opt.get("classifier_lr", lr),
             weight_decay=cfg_opt.get("classifier_weight_decay", wd)),
    ]
    kind = cfg_opt["type"].lower()
    if kind == "adam":
        return torch.optim.Adam(groups)
    if kind == "radam":
        return torch.optim.RAdam(groups)
    if kind == "nadam":
        return torch.optim.NAdam(groups, decoupled_weight_decay=True)
    if kind == "sgd":
        return torch.optim.SGD(groups)
    raise NotImplementedError(f"Unknown optimizer in config: {cfg_opt['type']}")


def make_scheduler(opt, lr_policy: dict):
    """utils.py:45-61."""
    if len(lr_policy) == 0:
        return None
    kind = lr_policy["type"]
    if kind == "step":
        return torch.optim.lr_scheduler.StepLR(opt, step_size=lr_policy["step_size"], gamma=lr_policy["gamma"])
    if kind == "multistep":
        return torch.optim.lr_scheduler.MultiStepLR(opt, milestones=lr_policy["steps"], gamma=lr_policy["gamma"])
    if kind == "cosine":
        return torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=lr_policy["n_epochs"])
    raise NotImplementedError(f"Learning rate policy {kind} not implemented.")


# ---------------------------------------------------------------- logger ----
class EpochLog:
    """logging.py:245-294 result-dict contract (python lists)."""

    def __init__(self, multi: bool):
        self.multi = multi
        self.reset()

    def reset(self):
        mk = (lambda: defaultdict(list)) if self.multi else list
        self.loss, self.conf, self.pred, self.gt = mk(), mk(), mk(), mk()
        self.images = None

    def add(self, pred, true, loss):
        if self.multi:
            for t in pred:
                self.gt[t].extend(true[t].tolist())
                self.conf[t].extend(pred[t].detach().softmax(-1, dtype=torch.float32).numpy().tolist())
                self.pred[t].extend(pred[t].detach().argmax(-1).tolist())
                self.loss[t].append(loss[t].item())
            self.loss["loss"].append(loss["loss"].item())
        else:
            self.gt.extend(true.tolist())
            self.conf.extend(pred.detach().softmax(-1, dtype=torch.float32).numpy().tolist())
            self.pred.extend(pred.detach().argmax(-1).tolist())
            self.loss.append(loss.item())

    def results(self):
        return {"running_loss": self.loss, "confidences": self.conf, "predictions": self.pred,
                "ground_truth": self.gt, "images": self.images}


# ---------------------------------------------------------------- engine ----
def train_epoch(model, loader, optimizer, scheduler, criterion, log: EpochLog, log_gradients=False):
    """engine.py:20-85 without the tqdm bar: zero_grad -> fwd -> loss -> bwd -> step -> log."""
    model.train()
    log.reset()
    grad_log = defaultdict(list)
    for img, target in loader:
        optimizer.zero_grad()
        preds = model(img)
        loss = criterion(preds, target)
        (loss["loss"] if isinstance(loss, dict) else loss).backward()
        optimizer.step()
        log.add(preds, target, loss)
        if log_gradients:
            total = 0
            for tag, p in model.named_parameters():
                if p.grad is not None:
                    g = p.grad.norm()
                    grad_log[f"Gradients/{tag}"].append(g)
                    total = total + g
            grad_log["Gradients/Total"].append(total)
        if log.images is None:
            log.images = img
    if scheduler is not None:
        scheduler.step()
    res = log.results()
    if log_gradients:
        res["metrics_grad_log"] = grad_log
    return res


@torch.no_grad()
def val_epoch(model, loader, criterion, log: EpochLog):
    """engine.py:88-117."""
    model.eval()
    log.reset()
    for img, target in loader:
        preds = model(img)
        loss = criterion(preds, target)
        log.add(preds, target, loss)
        if log.images is None:
            log.images = img
    return log.results()


def synthetic_batches(n_images: int, batch: int, n_classes, seed: int = 1234, hw: int = 224):
    """SURVEY §8(d) synthetic inputs: randn images, randint labels, one generator."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_images // batch):
        img = torch.randn(batch, 3, hw, hw, generator=g)
        if isinstance(n_classes, dict):
            tgt = {t: torch.randint(0, c, (batch,), generator=g) for t, c in n_classes.items()}
        else:
            tgt = torch.randint(0, n_classes, (batch,), generator=g)
        out.append((img, tgt))
    return out
