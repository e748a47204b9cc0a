#!/usr/bin/env python
"""Headline benchmark: images/sec of the full train step (engine.train_epoch's per-batch body) on synthetic
3x224x224 batches, ResNet-50 single-task, bs=256 per GPU, bf16 compute with fp32 master weights
(BASELINE.json configs[1]) — one process per GPU, RCCL gradient all-reduce over xGMI for N>1.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` = images/sec over all ranks with inputs resident in HBM;
`roofline` = the dominant kernel family timed with HIP events on its own stream (libnkbhip profiler);
`cpu_baseline` = the CPU oracle (torch fp32 restatement of the reference path) timed on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
import types
from pathlib import Path

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "nkb-classification_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The step uses three HIP streams (compute, weight gradients, gradient exchange) next to RCCL's own.  The runtime maps
# streams onto 4 hardware queues by default, and once RCCL has taken its share the weight-gradient stream lands on the
# compute stream's queue: everything serialises (measured, single-rank rehearsal of the exchange path: 26.8 ms/step with
# 4 queues, 21.9 with 8; 21.6 without the exchange).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic FLOPs per image of one train step (fwd + dgrad + wgrad MACs x2), SURVEY.md §8(d)
TRAIN_GFLOP_PER_IMG = {"resnet18": 10.645, "resnet50": 24.287, "vit_base_patch16_224": 105.147,
                       "unicom ViT-L/14": 485.4}   # SURVEY.md §8(d) algorithmic FLOPs (fwd + dgrad + wgrad)
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3, "fp8": 5000.0}   # dense MFMA peaks, MI355X_MICROARCH.md.  fp8: the GEMMs issue
# v_mfma_f32_16x16x128_f8f6f4 (twice the bf16 rate, the ~5 PF dense figure), so the fp8 run's GEMM family and its step are priced
# against 5 PFLOP/s although attention / the remaining bf16 work can only reach half of that
HBM_PEAK_GBS = 8000.0                          # HBM3E, MI355X_MICROARCH.md
# HBM-side traffic ON FILE per benchmarked workload (scripts/collect_profiles.sh -> scripts/pmc_traffic.py; regenerate after any
# kernel change).  key = (model, batch, dtype)
PMC_TRAFFIC_NAMES = {("resnet50", 256, "bf16"): "resnet50_bf16", ("vit_base_patch16_224", 256, "bf16"): "vit_b16_bf16",
                     ("unicom ViT-L/14", 128, "bf16"): "unicom_vit_l14_bf16", ("unicom ViT-L/14", 128, "fp8"): "unicom_vit_l14_fp8"}


def pmc_traffic_file(key):
    """the newest committed profiles/rNN_<workload>_pmc_traffic.json for this workload (round tags sort lexically), or None"""
    import glob
    stem = PMC_TRAFFIC_NAMES.get(key)
    if not stem:
        return None
    found = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f"r[0-9][0-9]*_{stem}_pmc_traffic.json")))
    return os.path.basename(found[-1]) if found else None
# profiler tag (api.hip kernel ids) -> families of scripts/pmc_traffic.py that hold the same launches
TRAFFIC_FAMILIES = {"conv_igemm": ("conv_igemm_fwd", "conv_igemm_bwd", "gemm8p_fwd", "gemm8p_bwd"),
                    "conv_wgrad": ("conv_wgrad", "wgrad8p", "wgrad8f", "wgrad3x3", "wgradr"), "bn_apply": ("bn_apply",),
                    "bn_bwd_apply": ("bn_bwd_apply",), "attn": ("attn_fwd", "attn_bwd", "attn_other"), "ln": ("layernorm",)}


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def pmc_traffic(args, kernel):
    """(HBM-side bytes per launch of `kernel`, HBM-side GB per step, file) from the committed rocprofv3 PMC passes over this same
    command (scripts/pmc_traffic.py: 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md prescribes).  A bench run cannot profile
    itself (counter collection needs its own rocprofv3 passes), so this is the measurement ON FILE for this workload — the line
    says so in roofline.traffic_source; a workload without a file reports null."""
    name = None if head_sizes(args) else pmc_traffic_file((args.model, args.batch, args.dtype))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name) if name else None
    if not (path and os.path.exists(path)):
        return None, None, None
    try:
        with open(path) as f:
            doc = json.load(f)
        ks = doc["kernels"]
        parts = [ks[k] for k in TRAFFIC_FAMILIES.get(kernel, (kernel,)) if k in ks]
        per_launch = round(sum(x["traffic_bytes_per_launch"] * x["launches"] for x in parts) / sum(x["launches"] for x in parts)) if parts else None
        return per_launch, round(doc["step_total_bytes"] / 1e9, 2), name
    except (KeyError, ValueError, ZeroDivisionError):
        return None, None, None


def usable_cpus(report: dict = None) -> int:
    """Host cores the CPU baseline runs on: min(affinity mask, cgroup quota, the pool's share of a GPU box's host).  The GPU
    boxes of this pool show all 256 host cores to every lease (no affinity mask, no cgroup quota) and hand each one-GPU lease a
    16-core share by rule, so the share is what bounds the count there; NKB_CPU_THREADS overrides it (any value, e.g. 256 on a
    box of one's own).  What was seen goes into `report` (printed, and into the line's cpu_baseline)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            quota = max(1, int(int(q) / int(period)))
    except Exception:
        pass
    n = min(aff, quota) if quota else aff
    env = os.environ.get("NKB_CPU_THREADS")
    share = 16                                       # rank 0 alone runs the baseline: one GPU lease = 16 host cores
    used = max(1, int(env)) if env else max(1, min(n, share))
    if report is not None:
        report.update(affinity=aff, cgroup_quota=quota, os_cpu_count=os.cpu_count(), pool_share=share,
                      override=int(env) if env else None)
    return used


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 10 warm-up, >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--classes", type=int, default=1000)
    ap.add_argument("--heads", default="", help="comma-separated class counts: the multi-task model (one Linear head per task, "
                    "FocalLoss gamma 1 summed over tasks — configs/multitask_config.py:146-176), e.g. 2,3,5,14 = BASELINE configs[3]")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8: bf16 autocast with the transformer blocks' Linear contractions in per-tensor-scaled fp8 "
                         "(BASELINE configs[4]); ResNet models have no fp8 path and run as bf16")
    ap.add_argument("--input", default="hbm", choices=["hbm", "host-uint8", "host-fp32"],
                    help="hbm: batch resident in HBM (the headline value). host-uint8: pinned uint8 HWC batches through "
                         "DeviceLoader (async H2D one batch ahead + on-GPU pad/flip/normalise). host-fp32: the reference's "
                         "way, a blocking fp32 .to(device) per step (engine.py:40).  The host variants are PCIe-inclusive "
                         "rates for DESIGN.md, never the headline.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-host-work", action="store_true", help="skip the batch-4 host-work measurement (profiler runs)")
    ap.add_argument("--cpu-batch", type=int, default=16)
    ap.add_argument("--cpu-steps", type=int, default=5)       # >= 5 timed steps
    return ap.parse_args()


def build(args, device):
    from nkb_classification.losses import get_loss
    from nkb_classification.model import get_model
    from nkb_classification.utils import get_optimizer
    torch.manual_seed(0)
    heads = head_sizes(args)
    cfg_model = dict(task="multi" if heads else "single", model=args.model, pretrained=False, backbone_dropout=0.0,
                     classifier_dropout=0.0, classifier_initialization="kaiming_normal_")
    classes = task_classes(heads) if heads else [str(i) for i in range(args.classes)]
    model = get_model(cfg_model, classes, device)
    # optimizer settings of configs/singletask_config.py:235-243 (NAdam, per-group lr / decoupled wd)
    opt = get_optimizer(model, dict(type="nadam", lr=1e-4, backbone_lr=1e-5, classifier_lr=1e-4, weight_decay=0.01,
                                    backbone_weight_decay=0.01, classifier_weight_decay=0.2))
    crit = get_loss(loss_config(heads), device)
    return model, opt, crit


def head_sizes(args):
    h = getattr(args, "heads", "") or ""
    return [int(v) for v in h.split(",") if v.strip()]


def task_classes(heads):
    return {f"task{i}": [str(c) for c in range(n)] for i, n in enumerate(heads)}


def loss_config(heads):
    # configs/multitask_config.py:176 / configs/singletask_config.py criterion
    return dict(task="multi", type="FocalLoss", gamma=1) if heads else dict(task="single", type="CrossEntropyLoss")


def make_targets(heads, classes, batch, generator, device):
    if heads:
        return {t: torch.randint(0, n, (batch,), generator=generator).to(device) for t, n in zip(task_classes(heads), heads)}
    return torch.randint(0, classes, (batch,), generator=generator).to(device)


def cpu_config0(threads):
    """BASELINE configs[0] exactly (the reference's own CPU-runnable case): ResNet-18, 2 classes, 64 synthetic
    224x224 images, bs 8, fp32, one epoch = 8 steps of the oracle's train step."""
    from oracle.torch_engine import Criterion, make_optimizer
    from oracle.torch_models import OracleClassifier
    torch.manual_seed(0)
    m = OracleClassifier(dict(model="resnet18", backbone_dropout=0.0, classifier_dropout=0.0), ["a", "b"])
    opt = make_optimizer(m, dict(type="nadam", lr=1e-4, backbone_lr=1e-5, classifier_lr=1e-4, weight_decay=0.01,
                                 backbone_weight_decay=0.01, classifier_weight_decay=0.2))
    crit = Criterion(dict(task="single", type="CrossEntropyLoss"))
    g = torch.Generator().manual_seed(1234)
    xs = torch.randn(64, 3, 224, 224, generator=g)
    ys = torch.randint(0, 2, (64,), generator=g)
    m.train()

    def epoch():
        for i in range(0, 64, 8):
            opt.zero_grad()
            crit(m(xs[i:i + 8]), ys[i:i + 8]).backward()
            opt.step()

    epoch()                                        # warm-up epoch
    t0 = time.perf_counter()
    epoch()
    dt = time.perf_counter() - t0
    return dict(value=round(64 / dt, 2), unit="images/sec", cores=threads,
                sample="configs[0]: resnet18 fp32, 2 classes, 64 images, bs=8, one timed epoch (8 steps) after one warm-up epoch")


def cpu_baseline(args):
    """Oracle (kind 'port'): torch-CPU fp32 restatement of the same train step, bounded sample."""
    from oracle.torch_engine import Criterion, make_optimizer
    from oracle.torch_models import OracleClassifier
    seen = {}
    threads = usable_cpus(seen)
    torch.set_num_threads(threads)
    log(f"cpu baseline on {threads} threads (affinity {seen['affinity']}, cgroup quota {seen['cgroup_quota']}, os.cpu_count() "
        f"{seen['os_cpu_count']}, pool share {seen['pool_share']}, NKB_CPU_THREADS {seen['override']})")
    torch.manual_seed(0)
    heads = head_sizes(args)
    m = OracleClassifier(dict(model=args.model, backbone_dropout=0.0, classifier_dropout=0.0, task="multi" if heads else "single"),
                         task_classes(heads) if heads else [str(i) for i in range(args.classes)])
    opt = make_optimizer(m, dict(type="nadam", lr=1e-4, backbone_lr=1e-5, classifier_lr=1e-4, weight_decay=0.01,
                                 backbone_weight_decay=0.01, classifier_weight_decay=0.2))
    crit = Criterion(loss_config(heads))
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(args.cpu_batch, 3, 224, 224, generator=g)
    y = make_targets(heads, args.classes, args.cpu_batch, g, "cpu")
    m.train()

    def step():
        opt.zero_grad()
        loss = crit(m(x), y)
        (loss["loss"] if heads else loss).backward()
        opt.step()

    step()  # warm-up
    log("cpu warm-up step done")
    t0 = time.perf_counter()
    for _ in range(args.cpu_steps):
        step()
        log(f"cpu step done at {time.perf_counter() - t0:.1f}s")
    dt = time.perf_counter() - t0
    out = dict(value=round(args.cpu_batch * args.cpu_steps / dt, 2), unit="images/sec", cores=torch.get_num_threads(),
               kind="port", sample=f"{args.model} fp32 train step, bs={args.cpu_batch}, {args.cpu_steps} timed steps "
                                  f"after 1 warm-up, torch {torch.__version__} CPU",
               cores_seen=seen)
    try:
        out["config0"] = cpu_config0(threads)
        log(f"cpu configs[0] (resnet18 bs 8): {out['config0']['value']} img/s")
    except Exception as e:                          # the headline baseline above stands on its own
        log(f"cpu configs[0] baseline failed: {e}")
    return out


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON: native libraries (the RCCL version banner, NCCL_DEBUG output) write to file
    # descriptor 1 behind Python's back, so fd 1 is pointed at stderr for the whole run and the line goes to a saved copy
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    # rehearsal knobs (not used by the driver): NKB_BENCH_DEVICE pins every rank to one GPU and NKB_DIST_BACKEND=gloo moves
    # the collectives through the host, so that the multi-rank control flow can be exercised on a one-GPU box
    dev_index = int(os.environ.get("NKB_BENCH_DEVICE", local))
    backend = os.environ.get("NKB_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    force_dist = os.environ.get("NKB_FORCE_REDUCER") == "1"     # rehearse the RCCL path with a single rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    dist_info = None
    if dist.is_initialized():
        # what the process group really is: every rank contributes a one, so `ranks_seen` is counted by the collective itself
        ones = torch.ones(1, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        dist_info = {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "world": dist.get_world_size(),
                     "ranks_seen": int(ones.item())}

    from nkb_classification import hip
    from nkb_classification.logging import softmax_argmax
    model, opt, crit = build(args, device)
    reducer = None
    if world > 1 or force_dist:
        from nkb_classification.parallel import GradReducer
        reducer = GradReducer(model, opt)
        if force_dist:
            reducer.world = 2        # take the multi-rank code path (bucketing, side stream, async work handles)
            opt.grad_scale = 1.0
            hip.rowres_reserve_cus(32)      # ... and the grid sizing a real multi-rank reducer asks for (parallel.GradReducer)

    g = torch.Generator().manual_seed(1234 + rank)
    img = torch.randn(args.batch, 3, 224, 224, generator=g).to(device)
    heads = head_sizes(args)
    tgt = make_targets(heads, args.classes, args.batch, g, device)
    if heads and args.input != "hbm":
        raise SystemExit("--heads runs with --input hbm only")
    amp = args.dtype in ("bf16", "fp8")
    model.fp8_linear = args.dtype == "fp8"
    model.train()

    def host_batches(n):
        """n batches in the chosen host format, cycling over two pinned buffers."""
        if args.input == "host-uint8":
            from nkb_classification.dataset import DeviceLoader
            raws = [torch.randint(0, 256, (args.batch, 224, 224, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(2)]
            tg = tgt.cpu().pin_memory()
            return iter(DeviceLoader([(raws[i & 1], None, tg) for i in range(n)], device, 224, hflip_p=0.5, seed=rank))
        host = [torch.randn(args.batch, 3, 224, 224, generator=g).pin_memory() for _ in range(2)]
        tg = tgt.cpu()
        return iter([(host[i & 1], tg) for i in range(n)])

    feed = None

    def step():
        nonlocal img, tgt
        x, t = img, tgt
        if feed is not None:
            x, t = next(feed)
            x, t = x.to(device), t.to(device)       # engine.py:40-41; a no-op for DeviceLoader batches
        opt.zero_grad()
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=amp):
            preds = model(x)
            loss = crit(preds, t)
        if heads:
            loss = loss["loss"]     # MultitaskCriterion: per-task losses + their sum (losses.py:97-151)
        loss.backward()
        opt.step()
        for p_ in (preds.values() if heads else (preds,)):
            softmax_argmax(p_)      # the logger's per-step by-products (device side, no host sync)
        return loss

    # first step packs the arena; broadcast rank-0 state afterwards so all ranks start identical
    log("model built; first step")
    step()
    torch.cuda.synchronize()
    log("first step done")
    if reducer is not None:
        reducer.broadcast_state(model.arena.flat_param, [b for b in model.buffers() if b.is_floating_point()])
        model.arena.mark_dirty()          # bf16 shadow / re-laid-out filters follow the received masters (as parallel.attach does)
    for _ in range(max(args.warmup - 1, 0)):
        step()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if True:
        # the interpreter's first full collection after model construction walks every module / tensor object (tens of
        # ms of host time, once); take it now instead of inside the first timed step — and BEFORE the barrier below, so
        # that no rank enters the timed region a collection ahead of another
        import gc
        gc.collect()
        gc.freeze()
    if args.input != "hbm":
        feed = host_batches(args.steps + 2)
        step(); step()                       # pipeline primed: the first copy is not hidden behind a step
    sync()                                   # barrier + synchronize: the last thing before the clock starts
    log("warm-up done; timing")
    t0 = time.perf_counter()
    step_marks = []
    for _ in range(args.steps):
        loss = step()
        step_marks.append(time.perf_counter())
    host_dt = time.perf_counter() - t0          # time to ENQUEUE the steps (host side only)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = float(loss.item())
    all_finite = math.isfinite(final_loss)
    if world > 1:
        # every rank has its own synthetic batch: a hazard can hit one rank only, and the others would then sit in the next
        # collective until it times out — agree on the verdict first (MIN over ranks), then all leave together
        flag = torch.tensor([1.0 if all_finite else 0.0], device=device if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        all_finite = bool(flag.item() > 0.5)
    if not all_finite:
        # synthetic data, random-init weights, lr 1e-4: a non-finite loss after a few dozen steps is a bug (a kernel, a stream
        # hazard), and a throughput measured on NaNs is not a measurement — fail instead of printing a line
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(f"bench.py: non-finite loss (this rank: {final_loss}) after the timed steps; refusing to report a throughput")
    log(f"timed region: {dt:.3f}s for {args.steps} steps")

    # host work per step: the enqueue loop above runs ahead until the HIP queue throttles it, so host_enqueue_ms_per_step is
    # queue back-pressure, not work.  The work itself is independent of the batch (same launches): time it at batch 4, where
    # the GPU finishes each step long before the host has enqueued the next.
    host_work = None
    if rank == 0 and args.input == "hbm" and world == 1 and not force_dist and not args.no_host_work:
        small_x = img[:4].clone()
        small_t = {k: v[:4].clone() for k, v in tgt.items()} if heads else tgt[:4].clone()
        keep = (img, tgt)
        img, tgt = small_x, small_t
        try:
            for _ in range(8):
                step()                               # sizes the batch-4 workspace, records its launch plans
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(20):
                step()
                torch.cuda.synchronize()             # every step starts with an empty queue: pure host time + a short GPU tail
            small_dt = (time.perf_counter() - t1) / 20
            t1 = time.perf_counter()
            for _ in range(20):
                step()
            host_work = (time.perf_counter() - t1) / 20
            torch.cuda.synchronize()
            log(f"host work at batch 4: {1e3 * host_work:.2f} ms/step enqueue ({1e3 * small_dt:.2f} ms/step end to end)")
        finally:
            img, tgt = keep
        for _ in range(6):
            step()                                   # back to the benchmark batch (workspace + plans re-established)
        torch.cuda.synchronize()

    feed = None
    roofline = None
    prof = {}
    if not args.no_roofline and rank != 0:
        # the profiled steps below exchange gradients like any other step, so every rank takes them; only rank 0 profiles
        for _ in range(min(args.steps, 5)):
            step()
        torch.cuda.synchronize()
    if rank == 0 and not args.no_roofline:
        # same step, re-run with one HIP-event pair per launch on the launch stream (perturbs wall time, so it is
        # kept out of the timed region above)
        nprof = min(args.steps, 5)
        # kernels that overlap on two streams stretch each other's event times, so the weight-gradient side stream is
        # folded back onto the main stream for these profiled steps (the timed region above keeps it)
        engines = list(getattr(getattr(model, "module", model), "_engines", {}).values())
        saved_overlap = [e.overlap_wgrad for e in engines]
        for e in engines:
            e.overlap_wgrad = False
        hip.prof_enable(True)
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        hip.prof_enable(False)
        for e, o in zip(engines, saved_overlap):
            e.overlap_wgrad = o
        prof = hip.prof_collect()
        # forward and data-gradient launches are the same kernel (conv_igemm_kernel<...>); the profiler only tags them
        # separately.  Rank kernels the way rocprofv3 --stats does, by kernel, so merge the two tags.
        merged = dict(prof)
        parts = [merged.pop(k) for k in ("conv_igemm_fwd", "conv_igemm_dgrad") if k in merged]
        if parts:
            merged["conv_igemm"] = {f: sum(x.get(f, 0) for x in parts) for f in ("ms", "launches", "work", "bytes")}
        cand = {k: v for k, v in merged.items() if v["work"] > 0 or v.get("bytes", 0) > 0}
        if cand:
            # dominant kernel family by device time; its bound is whichever floor is higher for the launches it made:
            # algorithmic FLOPs / dense MFMA peak or algorithmic bytes / HBM peak (both from MI355X_MICROARCH.md)
            name = max(cand, key=lambda k: cand[k]["ms"])
            v = cand[name]
            sec = v["ms"] * 1e-3
            peak_tf, peak_gbs = PEAK_TFLOPS[args.dtype], HBM_PEAK_GBS
            t_mfma, t_hbm = v["work"] / (peak_tf * 1e12), v.get("bytes", 0.0) / (peak_gbs * 1e9)
            total_ms = sum(x["ms"] for x in prof.values())
            traffic, step_traffic_gb, tfile = pmc_traffic(args, name)
            common = dict(traffic=traffic,
                          traffic_source=(f"profiles/{tfile}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                          "over this command, not this run") if traffic is not None else None,
                          kernel=name, launches_per_step=v["launches"] // nprof,
                          avg_launch_us=round(1e3 * v["ms"] / v["launches"], 2),
                          share_of_gpu_time=round(v["ms"] / total_ms, 3),
                          gflop_per_launch=round(v["work"] / v["launches"] / 1e9, 3),
                          mbyte_per_launch=round(v.get("bytes", 0.0) / v["launches"] / 1e6, 2),
                          mfma_frac=round(t_mfma / sec, 4), hbm_frac=round(t_hbm / sec, 4))
            if t_hbm >= t_mfma:
                ach = v["bytes"] / sec / 1e9
                roofline = dict(bound="hbm", achieved=round(ach, 1), peak=peak_gbs, unit="GB/s",
                                frac=round(ach / peak_gbs, 4), **common)
            else:
                ach = v["work"] / sec / 1e12
                roofline = dict(bound="mfma", achieved=round(ach, 2), peak=peak_tf, unit="TFLOP/s",
                                frac=round(ach / peak_tf, 4), **common)
    if world > 1:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        ips = args.batch * world * args.steps / dt
        gflop = TRAIN_GFLOP_PER_IMG.get(args.model)
        out = {
            "metric": "images/sec train step (3x224x224)",
            "value": round(ips, 1),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            # time the host spends inside the enqueue loop of the timed region: includes HIP-queue back-pressure when GPU-bound
            "host_enqueue_ms_per_step": round(1e3 * host_dt / args.steps, 3),
            # host WORK per step (Python + ctypes + runtime), measured at batch 4 where the queue never fills; same launches
            "host_work_ms_per_step": round(1e3 * host_work, 3) if host_work is not None else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": {"hbm": "synthetic randn images resident in HBM, random-init weights",
                     "host-uint8": "synthetic uint8 HWC batches in pinned host memory -> async H2D + on-GPU normalise/flip "
                                   "(PCIe-inclusive, not the headline), random-init weights",
                     "host-fp32": "synthetic fp32 NCHW batches in pinned host memory, blocking .to(device) per step "
                                  "(PCIe-inclusive, not the headline), random-init weights"}[args.input],
            "config": {"workload": (f"{args.model} multi-task train step, heads {args.heads} (FocalLoss gamma 1 per task, summed), "
                                    if heads else f"{args.model} single-task train step, {args.classes} classes, ") +
                                   f"bs={args.batch}/GPU, {args.dtype} compute + fp32 master weights, NAdam, 3x224x224",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "step_tflops": round(ips * gflop / 1e3, 1) if gflop else None,
            "step_mfma_frac": round(ips * gflop / 1e3 / PEAK_TFLOPS[args.dtype], 4) if gflop else None,
            "final_loss": round(final_loss, 4),
            # how the timed region is clocked (SURVEY 8(d) names hipEventRecord; with ONE device synchronisation at the end the host
            # clock brackets the same interval: barrier + synchronize, perf_counter, `steps` enqueued steps, synchronize (+ barrier),
            # perf_counter; MAX over ranks).  Per-kernel figures (roofline, kernel_ms_per_step) are HIP-event pairs on the launch stream.
            "timing": "host perf_counter around `steps` enqueued steps, bracketed by barrier + torch.cuda.synchronize() on both sides; max over ranks",
            # HBM-side GB per step from the same PMC passes the roofline's `traffic` comes from (null: no file for this workload)
            "step_traffic_gb": pmc_traffic(args, "conv_igemm")[1],
            # data parallel: what the process group really was (ranks counted by an all-reduce of ones)
            "dist": dist_info,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernel_ms_per_step": {k: round(v["ms"] / max(min(args.steps, 5), 1), 3) for k, v in
                                   sorted(prof.items(), key=lambda kv: -kv[1]["ms"])} if prof else None,
            # algorithmic bytes / event time, TB/s, for the launches that declare their bytes
            "kernel_tb_per_s": {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in
                                sorted(prof.items(), key=lambda kv: -kv[1]["ms"]) if v.get("bytes", 0) > 0} if prof else None,
        }
        print(json.dumps(out), file=real_stdout, flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
