"""Loss factories — drop-in for /root/reference/nkb_classification/losses.py.

`get_loss(cfg_loss, device)` (losses.py:154-176) returns a callable with the reference's contract: a scalar
tensor for single-task configs, a dict `{task: loss_t, ..., "loss": sum_t loss_t}` for multi-task ones
(losses.py:110-147).  CrossEntropyLoss(weight) and FocalLoss(alpha, gamma) are evaluated by one fused HIP
kernel pair (nkb_loss_forward / nkb_loss_backward) that also produces the softmax confidences and argmax the
epoch logger needs (logging.py:268-281), so no second pass over the logits is made.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Optional

import torch
from torch import Tensor, nn

from . import hip

DEFAULT_FOCAL_GAMMA = 2.0


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: Tensor, target: Tensor, kind: int, weight: Optional[Tensor], gamma: float,
                ignore_index: int, reduction: int = 0):
        B, C = logits.shape
        dev = logits.device
        ld = logits.stride(0)
        probs = torch.empty(B, C, device=dev, dtype=torch.float32)
        argmax = torch.empty(B, device=dev, dtype=torch.int32)
        rows = torch.empty(hip.load().nkb_loss_row_state_bytes(B), device=dev, dtype=torch.uint8)
        out2 = torch.empty(2, device=dev, dtype=torch.float32)
        hip.loss_forward(kind, logits, ld, target, B, C, weight, gamma, ignore_index, probs, C, argmax, rows, out2,
                         reduction)
        ctx.save_for_backward(probs, target, rows, out2)
        ctx.shape = (B, C)
        ctx.per_row = reduction == 2
        # "none": one loss per row (rows labelled ignore_index carry 0 here; FocalLoss.forward drops them afterwards)
        loss = rows.view(torch.float32).view(B, 3)[:, 0].clone() if ctx.per_row else out2[0].clone()
        ctx.mark_non_differentiable(probs, argmax)
        return loss, probs, argmax

    @staticmethod
    def backward(ctx, gout: Tensor, _gp=None, _ga=None):
        probs, target, rows, out2 = ctx.saved_tensors
        B, C = ctx.shape
        dl = torch.empty(B, C, device=probs.device, dtype=torch.float32)
        hip.loss_backward(probs, C, target, rows, out2, gout.contiguous().float(), B, C, dl, C, per_row=ctx.per_row)
        return dl, None, None, None, None, None, None


class _HipLoss(nn.Module):
    kind = 0

    def _class_weight(self) -> Optional[Tensor]:
        return None

    def _run(self, x: Tensor, y: Tensor, gamma: float, ignore_index: int, reduction: int = 0) -> Tensor:
        hip.require_device(x, type(self).__name__)
        if x.dim() != 2 or x.dtype != torch.float32 or x.stride(1) != 1:
            raise RuntimeError(f"{type(self).__name__}: expected fp32 logits [batch, classes], got {tuple(x.shape)} {x.dtype}")
        y = y.to(device=x.device, dtype=torch.int64).contiguous()
        w = self._class_weight()
        if w is not None and w.device != x.device:
            w = w.to(x.device)
        loss, probs, argmax = _LossFn.apply(x, y, self.kind, w, float(gamma), int(ignore_index), reduction)
        # by-products for the epoch logger (logging.py:268-281): softmax confidences and argmax of these logits
        x._nkb_side = (probs, argmax)
        return loss


class CrossEntropyLoss(_HipLoss):
    """nn.CrossEntropyLoss(weight) with 'mean' reduction (the only form losses.py:158 builds)."""
    kind = 0

    def __init__(self, weight: Optional[Tensor] = None):
        super().__init__()
        self.register_buffer("weight", weight)

    def _class_weight(self):
        return self.weight

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        return self._run(x, y, 0.0, -100)


class FocalLoss(_HipLoss):
    """Focal loss (https://arxiv.org/abs/1708.02002) with the reference's semantics (losses.py:10-94):
    rows labelled `ignore_index` are dropped, loss_i = -alpha[y] * (1 - p_y)^gamma * log p_y, 'mean' over kept rows."""
    kind = 1

    def __init__(self, alpha: Optional[Tensor] = None, gamma: float = DEFAULT_FOCAL_GAMMA, reduction: str = "mean",
                 ignore_index: int = -100):
        if reduction not in ("mean", "sum", "none"):
            raise ValueError('Reduction must be one of: "mean", "sum", "none".')
        super().__init__()
        self.register_buffer("alpha", alpha)
        self.gamma = gamma
        self.ignore_index = ignore_index
        self.reduction = reduction

    def __repr__(self):
        return (f"{type(self).__name__}(alpha={self.alpha!r}, gamma={self.gamma!r}, "
                f"ignore_index={self.ignore_index!r}, reduction={self.reduction!r})")

    def _class_weight(self):
        return self.alpha

    def forward(self, x: Tensor, y: Tensor) -> Tensor:
        if x.ndim > 2:
            c = x.shape[1]
            x = x.permute(0, *range(2, x.ndim), 1).reshape(-1, c)
            y = y.view(-1)
        if self.reduction == "mean":
            return self._run(x, y, self.gamma, self.ignore_index)
        # losses.py:66-70 / 89-94: "sum" adds the un-ignored rows' losses, "none" returns them row by row; an input whose
        # rows are all ignored yields a CPU scalar 0 (this branch reads the mask on the host, as the reference does)
        keep = y != self.ignore_index
        if not bool(keep.any()):
            return torch.tensor(0.0)
        if self.reduction == "sum":
            return self._run(x, y, self.gamma, self.ignore_index, reduction=1)
        return self._run(x, y, self.gamma, self.ignore_index, reduction=2)[keep.to(x.device)]


class MultitaskCriterion:
    """Applies one criterion per task and sums (losses.py:97-151)."""

    def __init__(self, criterion, device):
        self.criterion = criterion
        self.device = device
        self.criterion.to(device)

    def __call__(self, pred: dict, true: dict):
        assert pred.keys() == true.keys()
        total = 0
        out = defaultdict()
        for name in pred.keys():
            task_loss = self.criterion(pred[name], true[name].to(self.device))
            out[name] = task_loss
            total = total + task_loss
        out["loss"] = total
        return out


def get_loss(cfg_loss, device):
    if cfg_loss["type"] == "CrossEntropyLoss":
        weight = torch.tensor(cfg_loss["weight"], dtype=torch.float) if "weight" in cfg_loss else None
        loss = CrossEntropyLoss(weight).to(device)
    elif cfg_loss["type"] == "FocalLoss":
        alpha = torch.tensor(cfg_loss["alpha"], dtype=torch.float) if "alpha" in cfg_loss else None
        gamma = cfg_loss.get("gamma", DEFAULT_FOCAL_GAMMA)
        loss = FocalLoss(alpha, gamma).to(device)
    else:
        raise NotImplementedError(f'Unknown loss type in config: {cfg_loss["type"]}')
    if cfg_loss["task"] == "multi":
        return MultitaskCriterion(loss, device)
    return loss
