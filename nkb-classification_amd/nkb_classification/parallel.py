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

import os
from collections import defaultdict
from typing import List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    """bucket_dtype: "fp32" — ring all-reduce of fp32 arena slices (2 (W-1)/W x 4 B per parameter and rank); "bf16" — the mesh form
    for large models: every rank casts its bucket to bf16 and sends slice j to rank j (all-to-all), rank j adds the W slices in
    FP32, rounds the sum once and all ranks all-gather the bf16 sums (2 (W-1)/W x 2 B per parameter: half the bytes, and both
    phases are direct point-to-point transfers on the xGMI mesh instead of a ring); "auto" (default) picks bf16 above 100 M
    parameters — unicom ViT-L/14's 572 M parameters are 2.29 GB per step in fp32, which a ring over 153 GB/s links does not hide
    under a ~50 ms step.  Every rank ends up with bit-identical gradients in either mode."""

    def __init__(self, model, optimizer=None, process_group=None, bucket_bytes: int = 25 << 20, bucket_dtype: str = "auto",
                 reserve_cus: int = 32):
        if not dist.is_initialized():
            raise RuntimeError("GradReducer needs an initialised torch.distributed process group")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)      # (bench.py's one-rank rehearsal raises this to take the multi-rank path)
        self.group_world = self.world                        # what the collectives themselves see
        self.bucket_elems = max(1, bucket_bytes // 4)
        mode = os.environ.get("NKB_GRAD_BUCKET_DTYPE", bucket_dtype)
        if mode not in ("auto", "fp32", "bf16"):
            raise ValueError(f"bucket_dtype {mode!r}: expected auto, fp32 or bf16")
        total = getattr(getattr(model, "arena", None), "total", 0) or 0
        self.bf16_buckets = mode == "bf16" or (mode == "auto" and total > 100_000_000)
        if self.world > 1 and getattr(getattr(model, "arena", None), "device", None) is not None and model.arena.device.type == "cuda":
            # the collective's workgroups stay resident on a few CUs while backward runs: the one-workgroup-per-CU kernels of the
            # backward pass leave them room (a grid that does not fit runs a second round for a handful of workgroups)
            from . import hip
            hip.rowres_reserve_cus(reserve_cus)
        self._stage = {}                          # (bucket length, device) -> persistent bf16 staging buffers (_staging)
        self.stage_allocs = 0                     # how many of them were ever allocated (tests: constant after the first step)
        if self.bf16_buckets:
            # rank-local rounding before the sum and one more after it: not the fp32 all-reduce's numerics
            import logging
            logging.getLogger("nkb_classification").info(
                "GradReducer: bf16 gradient buckets (%s, %.0f M parameters): all-to-all + fp32 sum + all-gather", mode, total / 1e6)
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
                # cuts at 64-element boundaries of the arena: with range starts at parameter starts (64-aligned, runtime._ALIGN)
                # every bucket then begins 256-byte aligned, whatever the range's END is (a classifier bias of 10 elements) — the
                # cast / sum kernels of the bf16 exchange need 16-byte-aligned slices
                cut = max(a, (b - self.bucket_elems) // 64 * 64)
                todo.append((cut, b))
                b = cut
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
            a = max(lo, (b - self.bucket_elems) // 64 * 64)
            if self.bf16_buckets:
                self._exchange_bf16(flat[a:b])
            else:
                self._pending.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            b = a

    def _staging(self, n: int, device):
        """Persistent bf16 staging buffers of one bucket size: (send, recv, mine).  Buckets of a step run one after the other on
        the communication stream, and the sizes repeat from step to step, so each distinct size allocates ONCE (the padding tail of
        `send` is zeroed here and stays zero: the cast writes the first n elements, the all-gather writes sums of zeros behind
        them).  No allocator traffic inside the exchange after the first step — and nothing for a recorded launch plan to trip on."""
        key = (n, str(device))
        st = self._stage.get(key)
        if st is None:
            W = self.group_world
            chunk = -(-n // W)
            chunk += (-chunk) % 8                               # 16-byte-aligned slices for the sum kernel
            send = torch.zeros(W * chunk, device=device, dtype=torch.bfloat16)
            st = self._stage[key] = (send, torch.empty_like(send), torch.empty(chunk, device=device, dtype=torch.bfloat16), chunk)
            self.stage_allocs += 1
        return st

    def _exchange_bf16(self, g: torch.Tensor):
        """In-place sum over ranks of the fp32 slice g, moved as bf16 and accumulated in fp32 (see the class docstring).  Runs in
        stream order on the communication stream (cuda) or synchronously (cpu tensors: the gloo rehearsal)."""
        W, n = self.group_world, g.numel()
        send, recv, mine, chunk = self._staging(n, g.device)
        if g.is_cuda:
            from . import hip
            hip.wprep(hip.BF16, g, send, 1, 1, n, n, 0)         # fp32 -> bf16 (round to nearest even)
            dist.all_to_all_single(recv, send, group=self.group)
            hip.bucket_sum_bf16(recv, chunk, W, None, mine, chunk)
            dist.all_gather_into_tensor(send, mine, group=self.group)
            hip.bucket_sum_bf16(send, 0, 1, g, None, n)          # widen the gathered sums back into the arena
        else:
            send[:n] = g.to(torch.bfloat16)
            dist.all_to_all_single(recv, send, group=self.group)
            mine.copy_(recv.view(W, chunk).float().sum(0).to(torch.bfloat16))
            dist.all_gather_into_tensor(send, mine, group=self.group)
            g.copy_(send[:n].float())

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
