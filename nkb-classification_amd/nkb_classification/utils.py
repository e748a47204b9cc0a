"""Optimizer / scheduler factories and small config helpers.

Drop-in for /root/reference/nkb_classification/utils.py: `get_optimizer` (utils.py:10-42) keeps the two
parameter groups (backbone / classifier) with per-group lr and weight decay and the same optimizer
families, but the update itself is one fused HIP launch per group over the model's flat parameter arena
(nkb_optim_step) instead of torch's foreach kernels.  `get_scheduler` (utils.py:45-61) returns the stock
torch schedulers, which only touch `param_groups[i]["lr"]` on the host.
"""
from __future__ import annotations

import json
import math
import sys
from pathlib import Path

import numpy as np
import torch
from torch.optim import lr_scheduler

from . import hip

_KIND = {"adam": 0, "nadam": 1, "radam": 2, "sgd": 3}


def _step_scalars(kind: str, state: dict, *, lr: float, beta1: float, beta2: float, eps: float,
                  momentum_decay: float = 4e-3):
    """Advance the per-group step counter and return (kernel kind, (c0, c1, c2, c3)).

    The scalars are the step-dependent coefficients of torch.optim's single-tensor formulas, evaluated in
    double precision on the host exactly as torch does before it hands them to its kernels.
    """
    step = state["step"] = state.get("step", 0) + 1
    if kind == "sgd":
        return 3, (0.0, 0.0, 0.0, 0.0)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    if kind == "adam":
        return 0, (lr / bc1, math.sqrt(bc2), 0.0, 0.0)
    if kind == "nadam":
        mu = beta1 * (1.0 - 0.5 * (0.96 ** (step * momentum_decay)))
        mu_next = beta1 * (1.0 - 0.5 * (0.96 ** ((step + 1) * momentum_decay)))
        # torch keeps mu_product in a float32 state tensor (NAdam._init_group), so it is rounded to fp32 every step
        mu_product = state["mu_product"] = float(np.float32(state.get("mu_product", 1.0)) * np.float32(mu))
        return 1, (bc2, lr * (1.0 - mu) / (1.0 - mu_product), lr * mu_next / (1.0 - mu_product * mu_next), 0.0)
    if kind == "radam":
        rho_inf = 2.0 / (1.0 - beta2) - 1.0
        rho_t = rho_inf - 2.0 * step * (beta2 ** step) / bc2
        rect = 0.0
        if rho_t > 5.0:
            rect = math.sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t))
        return 2, (bc1, math.sqrt(bc2), rect, 0.0)
    raise NotImplementedError(kind)


class FusedOptimizer(torch.optim.Optimizer):
    """Adam / NAdam(decoupled) / RAdam / SGD with torch's defaults; the math runs in nkb_optim_step.

    When every parameter of a group is a view into the model's flat arena (see model.ParamArena) the group is
    updated by ONE launch over the contiguous range, which also refreshes the bf16 shadow weights; otherwise
    each parameter gets its own launch.  `grad_scale` (e.g. 1/world_size) is folded into the kernel.
    """

    def __init__(self, params, kind: str, arena=None):
        self.kind = kind
        # hyper-parameter keys and defaults of the torch optimizer the reference instantiates (utils.py:29-39)
        if kind == "sgd":
            defaults = dict(lr=1e-3, weight_decay=0.0, momentum=0, dampening=0, nesterov=False, maximize=False)
        else:
            defaults = dict(lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, maximize=False,
                            decoupled_weight_decay=(kind == "nadam"))
            if kind == "nadam":
                defaults["momentum_decay"] = 4e-3
            if kind == "adam":
                defaults["amsgrad"] = False
        super().__init__(params, defaults)
        self.arena = arena
        self.grad_scale = 1.0
        self._gstate = [dict() for _ in self.param_groups]

    def _arena_range(self, group):
        """(lo, hi) element range when the group's parameters and gradients tile one arena range, else None."""
        a = self.arena
        if a is None or not a.packed:
            return None
        ps = group["params"]
        if not ps or any(p.grad is None for p in ps):
            return None
        rng = a.range_of(ps)
        if rng is None:
            return None
        lo, hi = rng
        base_p, base_g = a.flat_param.data_ptr(), a.flat_grad.data_ptr()
        if ps[0].data_ptr() != base_p + 4 * a.offset_of(ps[0]) or ps[0].grad.data_ptr() != base_g + 4 * a.offset_of(ps[0]):
            return None
        if ps[-1].grad.data_ptr() != base_g + 4 * a.offset_of(ps[-1]):
            return None
        return lo, hi

    def rollback_last_step(self):
        """Undo the host-side bookkeeping of the most recent step() (its step counters and NAdam's mu_product): the
        gradient scaler found that the device skipped that step (amp.HipGradScaler), and torch does not count skipped steps."""
        if self._prev_gstate is not None:
            self._gstate = self._prev_gstate
            self._prev_gstate = None

    _prev_gstate = None

    @torch.no_grad()
    def step(self, closure=None, skip_flag=None):
        """skip_flag: device float; when it is non-zero at execution time the launches of this step do nothing."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._prev_gstate = [dict(g) for g in self._gstate]
        dev = next((p.device for g in self.param_groups for p in g["params"] if p.is_cuda), None)
        if dev is not None and dev != torch.device("cuda", torch.cuda.current_device()):
            with torch.cuda.device(dev):        # kernels go to the stream of the device that holds the parameters
                return self._step_impl(loss, skip_flag)
        return self._step_impl(loss, skip_flag)

    def _step_impl(self, loss, skip_flag):
        touched_arena = False
        shadowed = 0            # arena elements whose bf16 shadow this step's launches rewrote
        for group, gstate in zip(self.param_groups, self._gstate):
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            beta1, beta2 = group.get("betas", (0.0, 0.0))
            eps = group.get("eps", 0.0)
            lr, wd = float(group["lr"]), float(group["weight_decay"])
            kcode, sc = _step_scalars(self.kind, gstate, lr=lr, beta1=beta1, beta2=beta2, eps=eps,
                                      momentum_decay=group.get("momentum_decay", 4e-3))
            rng = self._arena_range(group)
            if rng is not None:
                lo, hi = rng
                a = self.arena
                m, v = a.moments()
                shadow = a.shadow[lo:hi] if a.shadow is not None else None
                hip.optim_step(kcode, a.flat_param[lo:hi], a.flat_grad[lo:hi], m[lo:hi], v[lo:hi], shadow, hi - lo,
                               lr, wd, beta1, beta2, eps, self.grad_scale, *sc, skip_flag=skip_flag)
                touched_arena = True
                if shadow is not None and len(live) == len(group["params"]):
                    shadowed += hi - lo
                continue
            for p in live:
                hip.require_device(p, "optimizer.step")
                st = self.state[p]
                if "exp_avg" not in st and self.kind != "sgd":
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g = p.grad
                if g.dtype != torch.float32 or p.dtype != torch.float32:
                    raise RuntimeError("FusedOptimizer: parameters and gradients must be fp32")
                if not (_dense(p) and _dense(g) and p.stride() == g.stride()):
                    raise RuntimeError("FusedOptimizer: parameter/gradient must be dense with equal strides")
                hip.optim_step(kcode, p, g, st.get("exp_avg"), st.get("exp_avg_sq"), None, p.numel(), lr, wd, beta1,
                               beta2, eps, self.grad_scale, *sc, skip_flag=skip_flag)
                if self.arena is not None and self.arena.owns(p):
                    touched_arena = True
        if touched_arena:
            self.arena.mark_dirty()
            if shadowed == self.arena.total:
                # every parameter went through a launch that also wrote its bf16 shadow: the engine's own refresh of the
                # whole shadow (one more pass over the arena per step) is redundant for this version
                self.arena.shadow_version = self.arena.version
        return loss


def _dense(t: torch.Tensor) -> bool:
    return t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last) or \
        t.numel() == t.untyped_storage().nbytes() // t.element_size()


def get_optimizer(model, cfg_optimizer):
    lr = cfg_optimizer.get("lr", 0.001)
    wd = cfg_optimizer.get("weight_decay", 0.0)
    groups = [
        {"params": list(model.emb_model.parameters()),
         "lr": cfg_optimizer.get("backbone_lr", lr),
         "weight_decay": cfg_optimizer.get("backbone_weight_decay", wd)},
        {"params": list(model.classifier.parameters()),
         "lr": cfg_optimizer.get("classifier_lr", lr),
         "weight_decay": cfg_optimizer.get("classifier_weight_decay", wd)},
    ]
    kind = cfg_optimizer["type"].lower()
    if kind == "sparse_adam":
        # utils.py:36 builds torch's SparseAdam; it needs sparse gradients, which no model of this package produces.
        return torch.optim.SparseAdam(groups)
    if kind not in _KIND:
        raise NotImplementedError(f'Unknown optimizer in config: {cfg_optimizer["type"]}')
    return FusedOptimizer(groups, kind, arena=getattr(model, "arena", None))


def get_scheduler(opt, lr_policy):
    if len(lr_policy) == 0:
        return None
    kind = lr_policy["type"]
    if kind == "step":
        return lr_scheduler.StepLR(opt, step_size=lr_policy["step_size"], gamma=lr_policy["gamma"])
    if kind == "multistep":
        return lr_scheduler.MultiStepLR(opt, milestones=lr_policy["steps"], gamma=lr_policy["gamma"])
    if kind == "cosine":
        return lr_scheduler.CosineAnnealingLR(opt, T_max=lr_policy["n_epochs"])
    raise NotImplementedError("Learning rate policy {} not implemented.".format(kind))


# ---- classes / config helpers (utils.py:64-105) --------------------------------------------
def save_classes(classes, save_path):
    if not isinstance(classes, (list, dict)):
        raise NotImplementedError(f"unknown classes config type {type(classes)}")
    with open(save_path, "w") as f:
        json.dump(classes, f)


def load_classes(classes):
    if isinstance(classes, (list, dict)):
        return classes
    if isinstance(classes, (str, Path)):
        with open(classes, "r") as f:
            return json.load(f)
    raise NotImplementedError(f"unknown classes config type {type(classes)}")


def get_classes_configs(classes):
    if isinstance(classes, list):
        c2i = {c: i for i, c in enumerate(classes)}
        return c2i, {i: c for c, i in c2i.items()}
    if isinstance(classes, dict):
        c2i = {t: {c: i for i, c in enumerate(cs)} for t, cs in classes.items()}
        return c2i, {t: {i: c for c, i in m.items()} for t, m in c2i.items()}
    raise NotImplementedError(f"unknown classes config type {type(classes)}")


def read_py_config(path):
    """Same contract as utils.py:101-105: returns the import statement the caller exec()s."""
    path = Path(path)
    sys.path.append(str(path.parent))
    return f"import {path.stem} as cfg"
