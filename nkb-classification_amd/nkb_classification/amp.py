"""Gradient scaler for the HIP engine — the `scaler` argument of engine.train_epoch
(/root/reference/train.py:37 `torch.cuda.amp.GradScaler(enabled=cfg.enable_gradient_scaler)`, used at
/root/reference/nkb_classification/engine.py:55-60: `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`).

Same constructor arguments, methods and state dict as torch's GradScaler; what differs is where the decision is taken.
torch reads `found_inf` back to the host inside `step()` — one blocking device->host copy per train step, which would stop
the enqueue loop from running ahead of the GPU.  Here
  * `nkb_grad_unscale_check` unscales the flat gradient arena in place and raises a device flag on inf / nan,
  * the fused optimizer launch (`nkb_optim_step(..., skip_flag)`) tests that flag on the device and does nothing when set,
  * `nkb_scaler_update` grows / backs off the scale on the device,
  * the host learns about a skipped step ONE STEP LATE, from a pinned copy of the flag that is normally complete by then, and
    only to keep the optimizer's step counters (bias corrections) exactly as torch would have them: a skipped step must not
    advance them, so the counters of the previous call are rolled back before the next one is evaluated.
With a torch optimizer (e.g. sparse_adam), or parameters outside the arena, there is no fused launch to hand the flag to: the
same HIP kernel then unscales every `p.grad` in place, ORing into the same device flag, and `step()` reads the flag once (the one
host sync of this slow path) to decide whether `optimizer.step()` runs — same scale, same growth tracker, same `update()`.
Under data parallelism the gradient exchange has finished before `step()` is called (model._backward_impl waits for the
reducer), so every rank tests identical reduced gradients and takes the same decision.
"""
from __future__ import annotations

from typing import Optional

import weakref

import torch

from . import hip


class HipGradScaler:
    def __init__(self, device="cuda", init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 enabled=True):
        if growth_factor <= 1.0 or not 0.0 < backoff_factor < 1.0:
            raise ValueError("growth_factor must be > 1 and backoff_factor in (0, 1)")
        self._enabled = bool(enabled) and torch.cuda.is_available()
        self._init_scale, self._growth_factor = float(init_scale), float(growth_factor)
        self._backoff_factor, self._growth_interval = float(backoff_factor), int(growth_interval)
        self._state: Optional[torch.Tensor] = None      # device [scale, found_inf, last_found_inf] fp32
        self._tracker: Optional[torch.Tensor] = None    # device int32 growth tracker
        self._host = None                               # pinned mirror of last_found_inf
        self._host_event = None
        self._last_opt = None                           # optimizer whose previous step may have been skipped
        self._unscaled = set()
        self._found = weakref.WeakKeyDictionary()       # optimizer -> its own inf / nan flag (non-fused path); dies with the optimizer
        self._init_growth_tracker = 0

    # ---- torch.amp.GradScaler surface ----------------------------------------------------------------------
    def is_enabled(self) -> bool:
        return self._enabled

    def _lazy_init(self, device):
        if self._state is None:
            self._state = torch.tensor([self._init_scale, 0.0, 0.0], device=device, dtype=torch.float32)
            self._tracker = torch.full((1,), self._init_growth_tracker, device=device, dtype=torch.int32)
            self._host = torch.zeros(1, dtype=torch.float32).pin_memory()

    def scale(self, outputs):
        if not self._enabled:
            return outputs
        if isinstance(outputs, torch.Tensor):
            self._lazy_init(outputs.device)
            return outputs * self._state[0].to(outputs.dtype)
        return type(outputs)(self.scale(o) for o in outputs)

    def get_scale(self) -> float:
        if not self._enabled:
            return 1.0
        self._settle()
        return self._init_scale if self._state is None else float(self._state[0].item())

    def _fused(self, optimizer):
        """One fused launch over the arena decides the skip on the device — only when EVERY parameter of the optimizer lives in
        the packed arena (anything else would be stepped with gradients that were neither unscaled nor checked)."""
        from .utils import FusedOptimizer
        if not (isinstance(optimizer, FusedOptimizer) and optimizer.arena is not None and optimizer.arena.packed):
            return False
        a = optimizer.arena
        return all(a.owns(p) for g in optimizer.param_groups for p in g["params"])

    def settle(self):
        """Public flush: make the optimizer's step counters reflect a possibly skipped LAST step (call before checkpoints)."""
        self._settle()

    def _settle(self):
        """Apply what the PREVIOUS step's flag says: a skipped step must not have advanced the optimizer's counters."""
        if self._host_event is None:
            return
        self._host_event.synchronize()                 # one step old: complete unless the host is a whole step ahead
        self._host_event = None
        if self._host[0] != 0.0 and self._last_opt is not None:
            self._last_opt.rollback_last_step()
        self._last_opt = None

    def unscale_(self, optimizer):
        if not self._enabled:
            return
        if id(optimizer) in self._unscaled:
            raise RuntimeError("unscale_() has already been called on this optimizer since the last update().")
        if not self._fused(optimizer):
            # per-parameter form: the same kernel over every gradient tensor.  The inf / nan flag is THIS optimizer's own (as in
            # torch.amp.GradScaler, which keeps found_inf per optimizer: an overflow in one optimizer's gradients must not skip
            # another one's step); it is folded into the shared flag only for update()'s growth / backoff decision
            own = None
            for group in optimizer.param_groups:
                for p in group["params"]:
                    g = p.grad
                    if g is None:
                        continue
                    if g.is_sparse:
                        if not g.is_coalesced():           # (torch coalesces before unscaling too: duplicates would overflow one by one)
                            g = g.coalesce()
                            p.grad = g
                        g = g._values()
                    if g.dtype != torch.float32 or not g.is_cuda:
                        raise RuntimeError("HipGradScaler: gradients must be fp32 tensors on a cuda device")
                    self._lazy_init(g.device)
                    if own is None:
                        own = self._found.get(optimizer)
                        if own is None:
                            own = self._found[optimizer] = torch.zeros(1, device=g.device)
                        own.zero_()
                    flat = g if g.is_contiguous() else None
                    if flat is None:
                        flat = g.contiguous()
                    with torch.cuda.device(g.device):
                        hip.grad_unscale_check(flat, flat.numel(), self._state[0:1], own)
                    if flat is not g:
                        g.copy_(flat)
            if own is not None:
                torch.maximum(self._state[1:2], own, out=self._state[1:2])
            self._unscaled.add(id(optimizer))
            return
        a = optimizer.arena
        self._lazy_init(a.flat_grad.device)
        with torch.cuda.device(a.flat_grad.device):
            hip.grad_unscale_check(a.flat_grad, a.total, self._state[0:1], self._state[1:2])
        self._unscaled.add(id(optimizer))

    def step(self, optimizer, *args, **kwargs):
        if not self._enabled:
            return optimizer.step(*args, **kwargs)
        self._settle()
        if id(optimizer) not in self._unscaled:
            self.unscale_(optimizer)
        if not self._fused(optimizer):
            if self._state is None:                     # nothing had a gradient
                return optimizer.step(*args, **kwargs)
            own = self._found.get(optimizer)
            if own is not None and float(own.item()) != 0.0:     # the slow path's one host read: skip, as torch's GradScaler.step does
                return None
            return optimizer.step(*args, **kwargs)
        out = optimizer.step(*args, skip_flag=self._state[1:2], **kwargs)
        self._last_opt = optimizer
        return out

    def update(self, new_scale=None):
        if not self._enabled:
            return
        if self._state is None:
            return
        if new_scale is not None:
            self._state[0] = float(new_scale)
        with torch.cuda.device(self._state.device):
            hip.scaler_update(self._state[0:1], self._tracker, self._state[1:2], self._state[2:3], self._growth_factor,
                              self._backoff_factor, self._growth_interval)
            self._host.copy_(self._state[2:3], non_blocking=True)
            self._host_event = torch.cuda.Event()
            self._host_event.record()
        for own in self._found.values():                # a flag belongs to ONE step: an optimizer without gradients next step
            own.zero_()                                 # must not inherit this step's overflow
        self._unscaled.clear()

    def state_dict(self):
        if not self._enabled:
            return {}
        self._settle()
        tracker = self._init_growth_tracker if self._tracker is None else int(self._tracker.item())
        return {"scale": self.get_scale(), "growth_factor": self._growth_factor, "backoff_factor": self._backoff_factor,
                "growth_interval": self._growth_interval, "_growth_tracker": tracker}

    def load_state_dict(self, state):
        if not self._enabled:
            return
        self._init_scale = float(state["scale"])
        self._growth_factor, self._backoff_factor = float(state["growth_factor"]), float(state["backoff_factor"])
        self._growth_interval = int(state["growth_interval"])
        self._init_growth_tracker = int(state["_growth_tracker"])
        if self._state is not None:
            self._state[0] = self._init_scale
            self._tracker.fill_(self._init_growth_tracker)


def GradScaler(device="cuda", **kwargs):
    """torch.amp.GradScaler(device, ...) spelling for call sites written against torch's class."""
    return HipGradScaler(device, **kwargs)
