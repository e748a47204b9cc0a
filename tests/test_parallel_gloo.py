"""world_size-2 rehearsal of the data-parallel path on CPU (gloo): bucketed all-reduce of flat-arena ranges
issued while 'backward' is still producing gradients, then the 1/world average through grad_scale."""
import os
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nkb_classification.parallel import GradReducer


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 10_000
        arena = types.SimpleNamespace(flat_grad=torch.zeros(n))
        model = types.SimpleNamespace(arena=arena, grad_ready_hook=None)
        calls = []
        opt = types.SimpleNamespace(grad_scale=1.0, step=lambda: calls.append(arena.flat_grad.clone()))
        red = GradReducer(model, opt, bucket_bytes=4 * 1500)       # several buckets per range
        assert model.grad_ready_hook is not None and opt.grad_scale == 0.5
        flat_param = torch.full((8,), float(rank + 1))
        red.broadcast_state(flat_param, [torch.zeros(0)])
        assert flat_param.tolist() == [1.0] * 8
        # backward produces the tail of the arena first, then the head
        arena.flat_grad[6000:] = torch.arange(6000, n, dtype=torch.float32) * (rank + 1)
        model.grad_ready_hook(6000, n)
        arena.flat_grad[:6000] = torch.arange(0, 6000, dtype=torch.float32) * (rank + 1)
        model.grad_ready_hook(0, 6000)
        opt.step()                                                   # wrapped: waits for every bucket first
        expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok = torch.equal(calls[0], expect) and not red._pending
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _gather_worker(rank, world, port, q):
    """Odd dataset length: 7 samples over 2 ranks.  The padded training shards hold 4 + 4 samples, one of them a
    repeat; the gathered epoch must hold each of the 7 exactly once.  Unpadded validation shards (4 + 3) pass through whole."""
    from nkb_classification.dataset import ShardedSampler
    from nkb_classification.parallel import gather_epoch_results
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 7
        ok = True
        for pad in (True, False):
            smp = ShardedSampler(n, rank, world, shuffle=True, seed=3, pad=pad)
            idx = list(iter(smp))
            ok &= len(idx) == len(smp) == (4 if pad else (4 if rank == 0 else 3)) and smp.real_len == (4 if rank == 0 else 3)
            res = dict(running_loss=[0.5] * 2, confidences=[[float(i)] for i in idx], predictions=list(idx), ground_truth=list(idx))
            out = gather_epoch_results(res, real_len=smp.real_len)
            ok &= sorted(out["ground_truth"]) == list(range(n)) and len(out["confidences"]) == n and len(out["running_loss"]) == 4
            multi = dict(running_loss={"a": [0.5], "loss": [0.5]}, confidences={"a": [[float(i)] for i in idx]},
                         predictions={"a": list(idx)}, ground_truth={"a": list(idx)})
            outm = gather_epoch_results(multi, real_len=smp.real_len)
            ok &= sorted(outm["ground_truth"]["a"]) == list(range(n))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_gathered_epoch_counts_every_sample_once_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _bf16_worker(rank, world, port, q):
    """bf16 bucket mode (all-to-all of bf16 slices, fp32 accumulation, all-gather of the sums) against the fp32 all-reduce on
    gradients that are exactly representable in bf16 together with their sums: the two modes must agree bit for bit, and an arena
    above 100 M parameters must select the bf16 mode by itself."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 10_007                                             # not a multiple of the world size or of the bucket
        outs = {}
        for mode in ("fp32", "bf16"):
            arena = types.SimpleNamespace(flat_grad=torch.zeros(n), total=n)
            model = types.SimpleNamespace(arena=arena, grad_ready_hook=None)
            opt = types.SimpleNamespace(grad_scale=1.0, step=lambda: None)
            red = GradReducer(model, opt, bucket_bytes=4 * 1500, bucket_dtype=mode)
            assert red.bf16_buckets == (mode == "bf16")
            arena.flat_grad[4000:] = (torch.arange(4000, n) % 61).float() * (rank + 1)
            model.grad_ready_hook(4000, n)
            arena.flat_grad[:4000] = (torch.arange(0, 4000) % 61).float() * (rank + 1)
            model.grad_ready_hook(0, 4000)
            opt.step()
            outs[mode] = arena.flat_grad.clone()
        expect = (torch.arange(n) % 61).float() * sum(r + 1 for r in range(world))
        ok = torch.equal(outs["fp32"], expect) and torch.equal(outs["bf16"], expect)
        big = types.SimpleNamespace(arena=types.SimpleNamespace(flat_grad=torch.zeros(8), total=572_328_448), grad_ready_hook=None)
        ok &= GradReducer(big).bf16_buckets and not GradReducer(types.SimpleNamespace(
            arena=types.SimpleNamespace(flat_grad=torch.zeros(8), total=25_557_032), grad_ready_hook=None)).bf16_buckets
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bf16_bucket_mode_matches_fp32_on_representable_gradients_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bf16_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _bf16_error_worker(rank, world, port, q):
    """bf16 bucket mode on REAL gradients (fp32 values that bf16 cannot hold, magnitudes over six decades, partly cancelling across
    the ranks): the error against the fp32 all-reduce is exactly two roundings — each rank's contribution to bf16 on the way out,
    the fp32 sum to bf16 on the way back — so per element |err| <= u (|g_0| + |g_1|) + u |sum| (1 + u) with bf16's unit roundoff u = 2^-8, and never more."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 50_021
        gen = torch.Generator().manual_seed(1234)
        base = torch.randn(world, n, generator=gen) * torch.logspace(-4, 2, n)[torch.randperm(n, generator=gen)]
        base[1, : n // 4] = -base[0, : n // 4] * (1.0 + 1e-3 * torch.randn(n // 4, generator=gen))     # near-cancelling pairs
        outs = {}
        for mode in ("fp32", "bf16"):
            arena = types.SimpleNamespace(flat_grad=torch.zeros(n), total=n)
            model = types.SimpleNamespace(arena=arena, grad_ready_hook=None)
            opt = types.SimpleNamespace(grad_scale=1.0, step=lambda: None)
            GradReducer(model, opt, bucket_bytes=4 * 7000, bucket_dtype=mode)
            arena.flat_grad[20000:] = base[rank, 20000:]
            model.grad_ready_hook(20000, n)
            arena.flat_grad[:20000] = base[rank, :20000]
            model.grad_ready_hook(0, 20000)
            opt.step()
            outs[mode] = arena.flat_grad.clone()
        exact = base.double().sum(0)
        ok = torch.allclose(outs["fp32"].double(), exact, rtol=1e-6, atol=1e-9)
        err = (outs["bf16"].double() - exact).abs()
        u = 2.0 ** -8
        bound = u * base.double().abs().sum(0) * (1 + u) + u * exact.abs() + 1e-30
        ok &= bool((err <= bound * 1.0001).all())
        rel = float(err.norm() / exact.norm())
        cos = float(torch.nn.functional.cosine_similarity(outs["bf16"].double(), exact, dim=0))
        ok &= rel < 4e-3 and cos > 1.0 - 1e-5
        ok &= bool((outs["bf16"] != outs["fp32"]).any())             # the inputs really were not representable
        q.put((rank, bool(ok), rel))
    finally:
        dist.destroy_process_group()


def test_bf16_bucket_mode_error_is_two_roundings_on_real_gradients_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bf16_error_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[:2] for r in results) == [(0, True), (1, True)], results


def test_reducer_requires_process_group():
    with pytest.raises(RuntimeError, match="process group"):
        GradReducer(types.SimpleNamespace(arena=None, grad_ready_hook=None))
