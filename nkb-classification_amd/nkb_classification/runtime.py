"""Device-memory plumbing of the HIP engine: the flat parameter arena, persistent activation workspace
and compute-dtype weight shadows.  No arithmetic happens here — every FLOP is in libnkbhip.

Layout decisions (MI355X, 288 GB HBM3E per GPU):
* all fp32 master parameters live in ONE flat buffer (`flat_param`), their gradients in a second flat buffer
  of the same shape (`flat_grad`), Adam moments in two more: the optimizer is one launch per parameter
  group, the data-parallel gradient exchange is a handful of large contiguous RCCL all-reduces, and
  zero_grad is one memset;
* a 4-D conv filter is stored [Cout][R][S][Cin] (what torch calls channels_last), which is the K-contiguous
  operand layout of the implicit-GEMM kernels; the nn.Parameter keeps its logical [Cout,Cin,R,S] shape so
  state_dict()/load_state_dict() stay timm-compatible;
* activations saved for backward are kept in a name-keyed workspace that is allocated once and reused
  every step (no allocator traffic inside the step, graph-capturable).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch
from torch import nn

from . import hip

_ALIGN = 64  # elements; keeps every parameter 256-byte aligned in fp32 and 128-byte aligned in bf16


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


class ParamArena:
    def __init__(self):
        self.packed = False
        self.device = None
        self.flat_param: Optional[torch.Tensor] = None
        self.flat_grad: Optional[torch.Tensor] = None
        self.shadow: Optional[torch.Tensor] = None  # bf16 copy of flat_param (allocated on first bf16 forward)
        self._m = self._v = None
        self._slots: Dict[int, Tuple[int, int]] = {}   # id(param) -> (offset, padded numel)
        self._params: List[nn.Parameter] = []
        self.version = 0          # bumped whenever parameter values may have changed
        self._seen_versions: List[int] = []
        self.total = 0

    # ---- construction ---------------------------------------------------------------
    def pack(self, blocks: Sequence[Sequence[nn.Parameter]], device):
        """blocks: lists of parameters; inside a block parameters are laid out back to back (only the block
        end is padded), blocks follow each other in the given order."""
        hip.require_device(torch.empty(0, device=device), "ParamArena.pack")
        off = 0
        slots = {}
        plist = []
        for block in blocks:
            for p in block:
                slots[id(p)] = [off, p.numel()]
                plist.append(p)
                off += p.numel()
            pad = _round_up(off, _ALIGN) - off
            if block:
                slots[id(block[-1])][1] += pad
            off += pad
        self.total = off
        flat = torch.zeros(off, device=device, dtype=torch.float32)
        grad = torch.zeros(off, device=device, dtype=torch.float32)
        with torch.no_grad():
            for p in plist:
                o, _ = slots[id(p)]
                view = self._view(flat, o, p)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = None
        self.flat_param, self.flat_grad = flat, grad
        self._slots = {k: (v[0], v[1]) for k, v in slots.items()}
        self._params = plist
        self.device = torch.device(device)
        self.shadow = None
        self._m = self._v = None
        self.packed = True
        self.mark_dirty()
        self._seen_versions = [p._version for p in plist]

    @staticmethod
    def _view(flat: torch.Tensor, off: int, p: torch.Tensor) -> torch.Tensor:
        n = p.numel()
        if p.dim() == 4:
            k, c, r, s = p.shape
            return flat[off:off + n].view(k, r, s, c).permute(0, 3, 1, 2)
        return flat[off:off + n].view(p.shape)

    # ---- queries ---------------------------------------------------------------------
    def owns(self, p) -> bool:
        return id(p) in self._slots

    def offset_of(self, p) -> int:
        return self._slots[id(p)][0]

    def range_of(self, params: Sequence[nn.Parameter]):
        if not params or any(id(p) not in self._slots for p in params):
            return None
        lo = min(self._slots[id(p)][0] for p in params)
        hi = max(self._slots[id(p)][0] + self._slots[id(p)][1] for p in params)
        if sum(self._slots[id(p)][1] for p in params) != hi - lo:
            return None
        return lo, hi

    def grad_view(self, p) -> torch.Tensor:
        return self._view(self.flat_grad, self.offset_of(p), p)

    def grad_flat(self, p) -> torch.Tensor:
        o = self.offset_of(p)
        return self.flat_grad[o:o + p.numel()]

    def param_flat(self, p) -> torch.Tensor:
        o = self.offset_of(p)
        return self.flat_param[o:o + p.numel()]

    def shadow_flat(self, p) -> torch.Tensor:
        o = self.offset_of(p)
        return self.shadow[o:o + p.numel()]

    def moments(self):
        if self._m is None:
            self._m = torch.zeros_like(self.flat_param)
            self._v = torch.zeros_like(self.flat_param)
        return self._m, self._v

    def still_packed(self) -> bool:
        """False when something (e.g. module.to(), load of foreign tensors) re-pointed a parameter elsewhere."""
        if not self.packed:
            return False
        base = self.flat_param.data_ptr()
        return all(p.data_ptr() == base + 4 * self._slots[id(p)][0] for p in self._params)

    # ---- change tracking --------------------------------------------------------------
    def mark_dirty(self):
        self.version += 1

    def poll_external_writes(self):
        """torch-side in-place writes (load_state_dict, init, torch optimizers) bump tensor._version."""
        cur = [p._version for p in self._params]
        if cur != self._seen_versions:
            self._seen_versions = cur
            self.mark_dirty()

    def ensure_shadow(self):
        if self.shadow is None:
            self.shadow = torch.empty(self.total, device=self.device, dtype=torch.bfloat16)

    # ---- gradients ---------------------------------------------------------------------
    def begin_backward(self, params_needing_grad: Sequence[nn.Parameter]):
        """Zero the flat gradient unless the caller is accumulating into our own views; then expose views."""
        ours = True
        fresh = True
        for p in params_needing_grad:
            if p.grad is not None:
                fresh = False
                if p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * self.offset_of(p):
                    ours = False
        if not ours:
            raise RuntimeError("HIP engine: .grad tensors were replaced by foreign tensors; call "
                               "optimizer.zero_grad() (set_to_none) before backward")
        if fresh:
            self.flat_grad.zero_()
        return fresh

    def publish_grads(self, params_needing_grad: Sequence[nn.Parameter]):
        for p in params_needing_grad:
            if p.grad is None:
                p.grad = self.grad_view(p)


# NKB_POISON_WS=1 (debugging): every fresh workspace buffer is filled with NaN bit patterns instead of being left as the allocator
# returned it, so a kernel that reads scratch it never wrote (an unwritten slab, padding behind a statistics block) turns the step
# non-finite at once instead of once in a few hundred steps (scripts/soak_determinism.py runs with it)
_POISON = os.environ.get("NKB_POISON_WS", "0") != "0"


def _poison(t: torch.Tensor) -> torch.Tensor:
    if _POISON and t.numel():
        if t.is_floating_point():
            t.fill_(float("nan"))
        else:
            t.view(torch.uint8).fill_(0xFF)
    return t


_GUARD_BYTES = 64 * 1024
_GUARDED: List[torch.Tensor] = []      # (poison mode) the flat allocations behind guarded buffers: [guard | payload | guard]


def _guarded(shape, dtype, device, zero=False) -> torch.Tensor:
    """Poison mode: the buffer sits between two 64 KB guard zones of NaN, so that a read a few rows past either end brings NaN into
    the step, and a WRITE past either end is found by guards_intact() afterwards."""
    n = 1
    for v in shape:
        n *= int(v)
    g = _GUARD_BYTES // torch.empty(0, dtype=dtype).element_size()
    flat = _poison(torch.empty(n + 2 * g, device=device, dtype=dtype))
    _GUARDED.append(flat)
    mid = flat[g:g + n].view(shape)
    if zero:
        mid.zero_()
    return mid


def guards_intact() -> bool:
    """Poison mode: every guard zone allocated so far still holds its fill (no kernel wrote outside its buffer)."""
    for flat in _GUARDED:
        g = _GUARD_BYTES // flat.element_size()
        for zone in (flat[:g], flat[-g:]):
            b = zone.view(torch.uint8)
            ok = torch.isnan(zone).all() if flat.is_floating_point() else (b == 0xFF).all()
            if not bool(ok):
                return False
    return True


def empty(*shape, **kw) -> torch.Tensor:
    """torch.empty for the engine's persistent shadows (padded weight layouts, folded filters): poisoned and guarded under NKB_POISON_WS."""
    if _POISON:
        return _guarded(tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape,
                        kw.get("dtype", torch.float32), kw.get("device"))
    return torch.empty(*shape, **kw)


class Workspace:
    """Name-keyed persistent device buffers (activations saved for backward, scratch, statistics)."""

    def __init__(self, device):
        self.device = device
        self._bufs: Dict[str, torch.Tensor] = {}
        self._allocs = 0          # bumped by every (re)allocation: recorded launch plans hold raw pointers into these buffers

    @property
    def generation(self):
        """What a recorded launch plan is valid for: this workspace's allocations AND the library's grid-sizing settings (a plan
        recorded under another nkb_rowres_reserve_cus replays launches whose partial-row / slab counts no longer match its buffers)."""
        return (self._allocs, hip.geometry_epoch())

    def get(self, name: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        shape = tuple(int(s) for s in shape)
        t = self._bufs.get(name)
        if t is None or t.shape != shape or t.dtype != dtype:
            if _POISON:
                t = _guarded(shape, dtype, self.device, zero)
            else:
                t = (torch.zeros if zero else torch.empty)(shape, device=self.device, dtype=dtype)
            self._bufs[name] = t
            self._allocs += 1
        return t

    def at_least(self, name: str, numel: int, dtype) -> torch.Tensor:
        """Flat scratch that only ever grows."""
        t = self._bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = _guarded((max(int(numel), 1),), dtype, self.device) if _POISON else torch.empty(max(int(numel), 1), device=self.device, dtype=dtype)
            self._bufs[name] = t
            self._allocs += 1
        return t

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._bufs.values())

    def clear(self):
        self._bufs.clear()
        self._allocs += 1
