"""Data-parallel gradient exchange for the HIP engine: one process per GPU, RCCL over xGMI through
torch.distributed (backend "nccl" is RCCL on ROCm).

The reference is single-device (train.py:98); this is new functionality required by BASELINE configs 4-5.
Because all gradients live in ONE flat fp32 arena laid out in forward order, backward completes the arena
from its tail to its head; as soon as a contiguous range is final the model calls `on_range_ready(lo, hi)`
and the range is all-reduced in large buckets on a side HIP stream while backward keeps running on the
compute stream.  The 1/world_size average is folded into the fused optimizer kernel (grad_scale).
BatchNorm statistics stay per-GPU (no SyncBN), as in standard DDP.
"""
from __future__ import annotations

from collections import defaultdict
from typing import List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, model, optimizer=None, process_group=None, bucket_bytes: int = 25 << 20):
        if not dist.is_initialized():
            raise RuntimeError("GradReducer needs an initialised torch.distributed process group")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._pending: List = []
        self._ready: List[List[int]] = []          # finished, not yet sent arena intervals [lo, hi), merged and sorted
        self._side_event = None
        self._stream: Optional[torch.cuda.Stream] = None
        model.grad_ready_hook = self.on_range_ready
        model.grad_done_hook = self.wait          # end of backward: nothing reads a gradient before the exchange is done
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
            self._wrap_step(optimizer)        # idempotent second wait for models that do not call grad_done_hook

    # -- start-of-training synchronisation: parameters and BN buffers from rank 0 --------------------
    def broadcast_state(self, flat_param: torch.Tensor, buffers=()):
        dist.broadcast(flat_param, 0, group=self.group)
        for b in buffers:
            if b.numel():
                dist.broadcast(b, 0, group=self.group)

    # -- called by the model while backward is still running --------------------------------------------
    def on_range_ready(self, lo: int, hi: int, side_event=None):
        """Gradients [lo, hi) of the flat arena are final once everything enqueued so far on the compute stream — and,
        when given, `side_event` of the weight-gradient stream — has run.  Finished ranges are merged and exchanged in
        buckets of at least `bucket_bytes` (a few large collectives instead of one per layer: RCCL ring time on xGMI is
        per-link bound, and every call costs host time that the enqueue loop does not have to spare); the rest goes out
        in wait().  The compute stream never waits here: only the communication stream does."""
        if self.world == 1 or hi <= lo:
            return
        if side_event is not None:
            self._side_event = side_event
        self._ready.append([lo, hi])
        self._ready.sort()
        merged = [self._ready[0]]
        for a, b in self._ready[1:]:
            if a <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], b)
            else:
                merged.append([a, b])
        self._ready = merged
        self._flush(final=False)

    def _flush(self, final: bool):
        keep = []
        todo = []
        for a, b in self._ready:
            # send whole buckets from the tail of the interval (produced first); keep the remainder for later merging
            while b - a >= self.bucket_elems:
                todo.append((b - self.bucket_elems, b))
                b -= self.bucket_elems
            if b > a:
                (todo if final else keep).append((a, b) if final else [a, b])
        self._ready = keep
        if not todo:
            return
        flat = self.model.arena.flat_grad
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record()                       # everything enqueued so far on the compute stream
            self._stream.wait_event(ev)
            if self._side_event is not None:
                self._stream.wait_event(self._side_event)
            with torch.cuda.stream(self._stream):
                for a, b in todo:
                    self._launch(flat, a, b)
        else:
            for a, b in todo:
                self._launch(flat, a, b)

    def _launch(self, flat, lo, hi):
        # walk from the tail: those gradients were produced first
        b = hi
        while b > lo:
            a = max(lo, b - self.bucket_elems)
            self._pending.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            b = a

    def wait(self):
        """Send what is left, then make the compute stream wait for every outstanding bucket (no host block on GPU)."""
        self._flush(final=True)
        self._side_event = None
        for w in self._pending:
            w.wait()
        self._pending.clear()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def _wrap_step(self, optimizer):
        inner = optimizer.step
        reducer = self

        def step(*a, **k):
            reducer.wait()
            return inner(*a, **k)

        optimizer.step = step


# ---- helpers for the config-driven entry point (train.py) ------------------------------------------------------------------
def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def attach(model, optimizer, device) -> GradReducer:
    """Make `model` a data-parallel replica: pack its parameter arena on `device`, copy rank 0's parameters, BatchNorm
    running statistics and `num_batches_tracked` to every rank, and hook the gradient exchange into backward."""
    device = torch.device(device)
    if not model.arena.still_packed() or model.arena.device != device:
        model._pack(device)                                       # parameters become views of one flat buffer
    reducer = GradReducer(model, optimizer)
    buffers = [b for name, b in model.named_buffers() if not name.endswith("num_batches_tracked")]
    nbt = getattr(model, "_nbt_flat", None)
    reducer.broadcast_state(model.arena.flat_param, buffers + ([nbt] if nbt is not None else []))
    model.arena.mark_dirty()                                      # shadows / folded filters follow the received values
    return reducer


def gather_epoch_results(results: dict, real_len=None) -> dict:
    """Concatenate the per-rank result lists of engine.train_epoch / val_epoch (logging.py:287-294) in rank order, so that
    metrics.compute_metrics sees the whole epoch on every rank.  Single-process runs pass through.
    real_len: how many of THIS rank's samples are real (ShardedSampler.real_len) — a padded training shard repeats the first
    indices of the permutation at its tail, and those repeats must not be counted twice in the epoch metrics or in the choice
    of best.pth; the per-sample lists (confidences, predictions, ground truth) are cut to it before the exchange."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return results
    keys = ("running_loss", "confidences", "predictions", "ground_truth")
    mine = {k: results[k] for k in keys}
    if real_len is not None:
        for k in ("confidences", "predictions", "ground_truth"):
            v = mine[k]
            mine[k] = {t: vals[:real_len] for t, vals in v.items()} if isinstance(v, dict) else v[:real_len]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, mine)
    out = dict(results)
    for k in keys:
        if isinstance(results[k], dict):
            merged = defaultdict(list)
            for part in parts:
                for task, vals in part[k].items():
                    merged[task].extend(vals)
            out[k] = merged
        else:
            out[k] = [v for part in parts for v in part[k]]
    return out
