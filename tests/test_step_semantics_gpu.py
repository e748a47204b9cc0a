"""Step-level semantics around the kernels: gradient scaler (engine.py:55-60), logger lists (logging.py:245-294), focal-loss
reductions (losses.py:89-94), label validation, and the data-parallel entry point (train.py under torch.distributed.run)."""
import json
import math
import os
import subprocess
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from nkb_classification import hip  # noqa: E402
from nkb_classification.amp import HipGradScaler  # noqa: E402
from nkb_classification.logging import BaseLogger  # noqa: E402
from nkb_classification.losses import FocalLoss, get_loss  # noqa: E402
from nkb_classification.model import get_model  # noqa: E402
from nkb_classification.utils import get_optimizer  # noqa: E402
from oracle import torch_engine as oe  # noqa: E402

DEV = "cuda:0"
ROOT = Path(__file__).resolve().parents[1]


def _tiny(seed=0, classes=("a", "b", "c")):
    cfg_model = dict(model="resnet_tiny_bottleneck", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    torch.manual_seed(seed)
    return get_model(cfg_model, list(classes), DEV)


# ---------------------------------------------------------------------------------------------- gradient scaler ----
@pytest.mark.parametrize("kind", ["nadam", "adam", "sgd"])
def test_grad_scaler_normal_overflow_and_recovery(kind):
    """A15: scale(loss).backward() / step / update.  Step 1 (finite) must equal the unscaled step of a twin model; step 2
    gets an inf injected into the gradients: parameters, moments and the optimizer's step counter stay put and the scale
    halves; step 3 (finite again) must equal the twin's SECOND step — i.e. the skipped step left no trace."""
    a, b = _tiny(), _tiny()
    b.load_state_dict(a.state_dict())
    cfg_opt = dict(type=kind, lr=1e-2, weight_decay=0.01)
    oa, ob = get_optimizer(a, cfg_opt), get_optimizer(b, cfg_opt)
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(4, 3, 64, 64, generator=g).to(DEV) for _ in range(3)]
    ys = [torch.randint(0, 3, (4,), generator=g).to(DEV) for _ in range(3)]
    scaler = HipGradScaler("cuda", init_scale=1024.0, growth_interval=2)
    a.train(); b.train()

    def twin_step(i):
        ob.zero_grad()
        crit(b(xs[i]), ys[i]).backward()
        ob.step()

    def scaled_step(i, poison=False):
        oa.zero_grad()
        scaler.scale(crit(a(xs[i]), ys[i])).backward()
        if poison:
            a.arena.flat_grad[a.arena.total // 2] = float("inf")
        scaler.step(oa)
        scaler.update()

    scaled_step(0)
    twin_step(0)
    torch.cuda.synchronize()
    torch.testing.assert_close(a.arena.flat_param, b.arena.flat_param, rtol=2e-5, atol=1e-6)
    assert scaler.get_scale() == 1024.0
    before = a.arena.flat_param.clone()
    m_before = a.arena.moments()[0].clone()
    scaled_step(1, poison=True)
    torch.cuda.synchronize()
    assert torch.equal(a.arena.flat_param, before) and torch.equal(a.arena.moments()[0], m_before)   # skipped on the device
    assert scaler.get_scale() == 512.0                                                                # backoff 0.5
    scaled_step(0 + 1)                           # same batch as the twin's second step
    twin_step(1)
    torch.cuda.synchronize()
    assert oa._gstate[0]["step"] == ob._gstate[0]["step"] == 2            # the skipped step was rolled back on the host
    torch.testing.assert_close(a.arena.flat_param, b.arena.flat_param, rtol=5e-5, atol=2e-6)
    sd = scaler.state_dict()
    assert sd["scale"] == 512.0 and sd["_growth_tracker"] == 1
    scaled_step(2)
    assert scaler.get_scale() == 1024.0          # growth after growth_interval = 2 clean steps


def test_grad_scaler_unscale_exposes_true_gradients():
    """unscale_() before step (the gradient-clipping idiom): gradients read afterwards are the unscaled ones, step() does not
    unscale twice, a second unscale_() raises like torch's."""
    a, b = _tiny(), _tiny()
    b.load_state_dict(a.state_dict())
    oa = get_optimizer(a, dict(type="sgd", lr=0.1))
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    x, y = torch.randn(4, 3, 64, 64).to(DEV), torch.tensor([0, 1, 2, 1]).to(DEV)
    a.train(); b.train()
    crit(b(x), y).backward()
    scaler = HipGradScaler("cuda", init_scale=4096.0)
    scaler.scale(crit(a(x), y)).backward()
    scaler.unscale_(oa)
    torch.cuda.synchronize()
    torch.testing.assert_close(a.arena.flat_grad, b.arena.flat_grad, rtol=1e-4, atol=1e-7)
    with pytest.raises(RuntimeError, match="already been called"):
        scaler.unscale_(oa)
    p0 = a.arena.flat_param.clone()
    scaler.step(oa)
    scaler.update()
    torch.testing.assert_close(a.arena.flat_param, p0 - 0.1 * b.arena.flat_grad, rtol=1e-4, atol=1e-6)
    assert HipGradScaler(enabled=False).scale(x) is x and HipGradScaler(enabled=False).get_scale() == 1.0


def test_grad_scaler_tracks_torch_amp_gradscaler_step_by_step():
    """A15 against the class the reference instantiates (/root/reference/train.py:37, used at engine.py:55-60): the SAME inf pattern is
    driven through torch.amp.GradScaler on the oracle module (torch kernels, fp32) and through HipGradScaler on the HIP model; after
    every step the scale, the growth tracker, the skip decision and the parameters must agree."""
    from oracle.torch_models import OracleClassifier
    cfg_model = dict(model="resnet_tiny_bottleneck", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    torch.manual_seed(0)
    ref = OracleClassifier(cfg_model, ["a", "b", "c"]).to(DEV)
    model = get_model(dict(cfg_model), ["a", "b", "c"], DEV)
    model.load_state_dict(ref.state_dict())
    cfg_opt = dict(type="sgd", lr=0.05, weight_decay=0.0)       # (updates linear in the gradients: parameters stay comparable)
    o_hip = get_optimizer(model, cfg_opt)
    o_ref = torch.optim.SGD(ref.parameters(), lr=0.05)
    kw = dict(init_scale=256.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2)
    s_hip, s_ref = HipGradScaler("cuda", **kw), torch.amp.GradScaler("cuda", **kw)
    g = torch.Generator().manual_seed(3)
    poison = [False, True, False, False, True, True, False, False, False]
    ref.train(); model.train()
    init = {n: p.detach().clone() for n, p in ref.named_parameters()}
    for i, bad in enumerate(poison):
        x = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
        y = torch.randint(0, 3, (4,), generator=g).to(DEV)
        o_ref.zero_grad()
        s_ref.scale(torch.nn.functional.cross_entropy(ref(x), y)).backward()
        if bad:
            next(ref.parameters()).grad.view(-1)[0] = float("inf")
        before_ref = [p.detach().clone() for p in ref.parameters()]
        s_ref.step(o_ref)
        s_ref.update()
        skipped_ref = all(torch.equal(a, b.detach()) for a, b in zip(before_ref, ref.parameters()))
        o_hip.zero_grad()
        s_hip.scale(torch.nn.functional.cross_entropy(model(x), y)).backward()
        if bad:
            model.arena.flat_grad[0] = float("inf")
        before_hip = model.arena.flat_param.clone()
        s_hip.step(o_hip)
        s_hip.update()
        torch.cuda.synchronize()
        skipped_hip = torch.equal(before_hip, model.arena.flat_param)
        assert skipped_hip == skipped_ref == bad, (i, skipped_hip, skipped_ref)
        assert s_hip.get_scale() == s_ref.get_scale(), (i, s_hip.get_scale(), s_ref.get_scale())
        assert s_hip.state_dict()["_growth_tracker"] == s_ref.state_dict()["_growth_tracker"], i
    # the parameters moved the same way: whole-model update vectors agree (BatchNorm over 4 images amplifies the kernels' fp32
    # summation-order differences in individual small tensors, e.g. the stem filter, so the comparison is on the whole update)
    sd = {k: v for k, v in model.state_dict().items()}
    d_hip = torch.cat([(sd[n].float() - init[n]).flatten() for n, _ in ref.named_parameters()])
    d_ref = torch.cat([(p.detach() - init[n]).flatten() for n, p in ref.named_parameters()])
    cos = torch.nn.functional.cosine_similarity(d_hip, d_ref, dim=0).item()
    assert cos > 0.98 and abs(d_hip.norm().item() / d_ref.norm().item() - 1.0) < 0.05, (cos, d_hip.norm().item(), d_ref.norm().item())


def test_grad_scaler_with_a_torch_optimizer_and_outside_parameters():
    """an optimizer that is not the fused one (here torch's SGD over the model's parameters plus one parameter that lives
    outside the arena) goes through the per-parameter unscale + one host read of the flag: finite step applied with UNSCALED
    gradients, poisoned step skipped, scale backs off — with the same scale / tracker state as the fused path."""
    a, b = _tiny(), _tiny()
    b.load_state_dict(a.state_dict())
    extra_a, extra_b = (torch.nn.Parameter(torch.ones(5, device=DEV)) for _ in range(2))
    oa = torch.optim.SGD(list(a.parameters()) + [extra_a], lr=0.1)
    ob = torch.optim.SGD(list(b.parameters()) + [extra_b], lr=0.1)
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    x, y = torch.randn(4, 3, 64, 64).to(DEV), torch.tensor([0, 1, 2, 1]).to(DEV)
    a.train(); b.train()
    scaler = HipGradScaler("cuda", init_scale=2048.0, growth_interval=100)
    assert not scaler._fused(oa)
    ob.zero_grad()
    (crit(b(x), y) + extra_b.pow(2).sum()).backward()
    ob.step()
    oa.zero_grad()
    scaler.scale(crit(a(x), y) + extra_a.pow(2).sum()).backward()
    scaler.step(oa)
    scaler.update()
    torch.cuda.synchronize()
    torch.testing.assert_close(a.arena.flat_param, b.arena.flat_param, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(extra_a.detach(), extra_b.detach(), rtol=1e-5, atol=1e-7)
    assert scaler.get_scale() == 2048.0
    keep, keep_extra = a.arena.flat_param.clone(), extra_a.detach().clone()
    oa.zero_grad()
    scaler.scale(crit(a(x), y) + extra_a.pow(2).sum()).backward()
    extra_a.grad[2] = float("nan")                                # the overflow sits in the parameter OUTSIDE the arena
    scaler.step(oa)
    scaler.update()
    torch.cuda.synchronize()
    assert torch.equal(a.arena.flat_param, keep) and torch.equal(extra_a.detach(), keep_extra)
    assert scaler.get_scale() == 1024.0 and scaler.state_dict()["_growth_tracker"] == 0


# ------------------------------------------------------------------------------------------------------ logger ----
def test_logger_lists_match_reference_golden(golden):
    """A16 / G5: the device-side logger returns the lists the reference's BaseLogger produced for the same inputs."""
    g5 = golden("g5_logger")
    single = g5["single"]
    lg = BaseLogger(types.SimpleNamespace(task="single"), list("abcdefg"))
    lg.init_iter_logs()
    for b in single["batches"]:
        lg.log_iter(torch.tensor(b["preds"]).to(DEV), torch.tensor(b["true"]).to(DEV), torch.tensor(b["loss"]).to(DEV))
    lg.log_images_if_needed(torch.zeros(1, 3, 2, 2, device=DEV))
    res = lg.get_epoch_results()
    np.testing.assert_allclose(res["confidences"], single["confidences"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(res["running_loss"], single["running_loss"], rtol=1e-7)
    assert res["predictions"] == single["predictions"] and res["ground_truth"] == single["ground_truth"]
    assert list(res["images"].shape) == single["images_shape"] and res["images"].device.type == "cpu"
    multi = g5["multi"]
    lg = BaseLogger(types.SimpleNamespace(task="multi"), multi["classes"])      # constructor works (logging.py:243 does not)
    assert lg.target_names == sorted(multi["classes"])
    lg.init_iter_logs()
    for b in multi["batches"]:
        lg.log_iter({t: torch.tensor(v).to(DEV) for t, v in b["preds"].items()},
                    {t: torch.tensor(v).to(DEV) for t, v in b["true"].items()},
                    {t: torch.tensor(v).to(DEV) for t, v in b["loss"].items()})
    res = lg.get_epoch_results()
    assert set(res["running_loss"]) == {"shape", "color", "loss"}
    for t in multi["classes"]:
        np.testing.assert_allclose(res["confidences"][t], multi["confidences"][t], rtol=2e-6, atol=1e-9)
        assert res["predictions"][t] == multi["predictions"][t] and res["ground_truth"][t] == multi["ground_truth"][t]
    for t in multi["running_loss"]:
        np.testing.assert_allclose(res["running_loss"][t], multi["running_loss"][t], rtol=1e-7)


# -------------------------------------------------------------------------------------------------- focal loss ----
@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
@pytest.mark.parametrize("gamma", [2.0, 0.5])
def test_focal_loss_reductions_and_gradients(reduction, gamma):
    """losses.py:59-94 for all three reductions, with ignored rows and class weights, values and d/dlogits vs the oracle."""
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(37, 6, generator=g) * 2).requires_grad_(True)
    y = torch.randint(0, 6, (37,), generator=g)
    y[[3, 17, 30]] = -100
    alpha = torch.rand(6, generator=g) + 0.5
    ref = oe.focal_loss(x, y, alpha=alpha, gamma=gamma, reduction=reduction)
    w = torch.randn(ref.shape, generator=g) if reduction == "none" else torch.tensor(1.7)
    (ref * w).sum().backward()
    xd = x.detach().to(DEV).requires_grad_(True)
    out = FocalLoss(alpha.to(DEV), gamma, reduction=reduction).to(DEV)(xd, y.to(DEV))
    assert out.shape == ref.shape
    (out * w.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(xd.grad.cpu(), x.grad, rtol=2e-4, atol=1e-6)
    # every row ignored: a CPU scalar zero in every reduction (losses.py:69-70)
    z = FocalLoss(None, gamma, reduction=reduction)(xd.detach(), torch.full((37,), -100, device=DEV))
    assert z.item() == 0.0


def test_focal_loss_saturated_row_small_gamma_has_finite_gradient():
    """0 < gamma < 1 and p_target == 1: (1-p)^(gamma-1) * log p is 0 * inf when evaluated literally."""
    x = torch.tensor([[80.0, -80.0, -80.0], [0.3, 0.1, -0.2]], device=DEV, requires_grad=True)
    loss = FocalLoss(None, 0.5)(x, torch.tensor([0, 1], device=DEV))
    loss.backward()
    assert torch.isfinite(loss).item() and torch.isfinite(x.grad).all().item() and x.grad[0].abs().max().item() == 0.0


def test_out_of_range_label_poisons_the_loss():
    """torch asserts on the device for a label outside [0, C); the sync-free kernel cannot, so the loss becomes NaN instead
    of silently training on the remaining rows."""
    x = torch.randn(5, 4, device=DEV, requires_grad=True)
    for crit in (get_loss(dict(task="single", type="CrossEntropyLoss"), DEV), FocalLoss(None, 2.0)):
        assert torch.isnan(crit(x, torch.tensor([0, 1, 7, 2, 3], device=DEV))).item()
        assert torch.isfinite(crit(x, torch.tensor([0, 1, 3, 2, 3], device=DEV))).item()


# ------------------------------------------------------------------------------------- data-parallel entry point ----
def _launch_train(tmp_path, nproc, extra_env=None, edits=()):
    root = ROOT / "nkb-classification_amd"
    cfg = (root / "configs" / "synthetic_singletask_config.py").read_text().replace(
        '"runs/synthetic_single"', repr(str(tmp_path / "exp"))).replace(
        "enable_gradient_scaler = False", "enable_gradient_scaler = True")
    for a, b in edits:
        assert a in cfg
        cfg = cfg.replace(a, b)
    assert "enable_gradient_scaler = True" in cfg
    (tmp_path / "cfg_ddp.py").write_text(cfg)
    env = dict(os.environ, NKB_DDP_BACKEND="gloo", NKB_DDP_ONE_GPU="1", NKB_DUMP_PARAMS=str(tmp_path), **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(29900 + os.getpid() % 500), str(ROOT / "tests" / "ddp_train_probe.py"), "-cfg",
           str(tmp_path / "cfg_ddp.py")]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)


def test_train_py_under_torchrun_two_ranks(tmp_path):
    """train.py launched as the driver launches bench.py (torch.distributed.run, 2 ranks; gloo transport, both ranks on the
    one GPU of this box): no manual seeding by the caller, rank 0's initial state is broadcast, each rank trains on its own
    shard with the gradient scaler ENABLED (the reference configs ship enable_gradient_scaler=True), and afterwards
      * both ranks hold bit-identical parameters and BatchNorm buffers,
      * exactly one run directory exists, written by rank 0, with one metrics row per epoch,
      * the metrics of an epoch cover the whole dataset (gathered from both ranks), not one shard."""
    r = _launch_train(tmp_path, 2)
    assert r.returncode == 0, r.stderr[-3000:]
    exp = tmp_path / "exp"
    assert exp.exists() and not (tmp_path / "exp1").exists()
    assert (exp / "weights" / "last.pth").exists() and (exp / "classes.json").exists()
    lines = (exp / "metrics.csv").read_text().strip().splitlines()
    assert len(lines) == 3 and lines[0].split("\t")[0] == "Epoch"
    d0, d1 = (torch.load(tmp_path / f"params_rank{r}.pt") for r in (0, 1))
    assert torch.equal(d0["flat_param"], d1["flat_param"])
    for k in d0["buffers"]:
        if "running" in k:
            continue        # BatchNorm statistics stay per GPU (no SyncBN), as in standard DDP
        assert torch.equal(d0["buffers"][k], d1["buffers"][k]), k
    assert d0["n_train"] == d1["n_train"] == d0["dataset_len"] > d0["shard_len"]     # gathered epoch covers every image
    sd = torch.load(exp / "weights" / "last.pth", map_location="cpu")
    torch.testing.assert_close(sd["classifier.1.weight"], d1["head_weight"])          # rank 1 agrees with rank 0's checkpoint


def test_bench_py_two_ranks_under_torchrun_prints_one_valid_line():
    """bench.py exactly as the driver launches it for N = 2 (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2),
    with the one-GPU rehearsal knobs (gloo transport, both ranks on this box's GPU): rank 0 prints ONE JSON line with n_gpus == 2,
    global_batch == 512, weak scaling, and a `dist` object whose ranks_seen was counted by an all-reduce over the process group."""
    import json
    env = dict(os.environ, NKB_DIST_BACKEND="gloo", NKB_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(30400 + os.getpid() % 500), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-host-work", "--no-roofline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 512 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["dist"] == {"backend": "gloo", "world": 2, "ranks_seen": 2}
    assert d["value"] > 0 and abs(d["value"] - 512 * 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]


def test_bench_py_host_uint8_input_keeps_its_labels_across_streams():
    """bench.py --input host-uint8: batches arrive through dataset.DeviceLoader (H2D on a copy stream one batch ahead) and the steps
    are enqueued without any host synchronisation.  The targets are allocated on the copy stream and read by the loss kernel on the
    compute stream; until round 3 they were not record_stream()-ed, so the allocator gave their block to the next staged batch's
    flip flags while the loss kernel was still queued: out-of-range classes, a NaN loss in every committed host-uint8 line.  The
    run must end with a finite loss (bench.py itself now refuses to print a line otherwise)."""
    import json
    cmd = [sys.executable, str(ROOT / "bench.py"), "--input", "host-uint8", "--steps", "30", "--warmup", "4", "--no-cpu-baseline",
           "--no-host-work", "--no-roofline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert math.isfinite(d["final_loss"]) and 3.0 < d["final_loss"] < 8.0, d["final_loss"]


def test_train_py_two_ranks_odd_dataset_counts_every_image_once(tmp_path):
    """with len(dataset) % world != 0 the padded training shard repeats an index and a padded validation shard would
    too — epoch metrics and the choice of best.pth must still be computed over each image exactly once (255 train / 127 val
    images on 2 ranks: shards of 128 + 128 with one repeat, validation shards of 64 + 63 unpadded)."""
    r = _launch_train(tmp_path, 2, edits=(('"n_images": 256', '"n_images": 255'), ('"n_images": 128', '"n_images": 127')))
    assert r.returncode == 0, r.stderr[-3000:]
    d0, d1 = (torch.load(tmp_path / f"params_rank{k}.pt") for k in (0, 1))
    assert d0["dataset_len"] == 255 and d0["val_len"] == 127
    assert d0["n_train"] == d1["n_train"] == 255 and d0["n_val"] == d1["n_val"] == 127
    assert d0["shard_len"] == d1["shard_len"] == 128
    assert torch.equal(d0["flat_param"], d1["flat_param"])


def _scaler_ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nkb_classification import parallel
        model = _tiny(seed=rank)                        # different initial weights on purpose: attach() must fix that
        opt = get_optimizer(model, dict(type="nadam", lr=1e-3))
        parallel.attach(model, opt, DEV)
        crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
        scaler = HipGradScaler("cuda", init_scale=256.0)
        g = torch.Generator().manual_seed(50 + rank)
        model.train()
        trace = []
        for step in range(3):
            x, y = torch.randn(4, 3, 64, 64, generator=g).to(DEV), torch.randint(0, 3, (4,), generator=g).to(DEV)
            opt.zero_grad()
            loss = crit(model(x), y)
            if step == 1 and rank == 1:
                loss = loss * float("inf")                 # overflow on ONE rank only
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            torch.cuda.synchronize()
            trace.append((scaler.get_scale(), float(model.arena.flat_param.double().sum())))
        q.put((rank, trace, opt._gstate[0]["step"]))
    finally:
        dist.destroy_process_group()


def test_grad_scaler_with_reducer_overflow_on_one_rank():
    """the exchange finishes before anything reads the gradients, so an overflow on one rank is seen by
    both (same skip decision, same scale, same parameters, no stranded collective)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 500)
    procs = [ctx.Process(target=_scaler_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, t0, s0), (_, t1, s1) = res
    assert t0 == t1                                       # identical scale and parameter checksum after every step
    assert [s for s, _ in t0] == [256.0, 128.0, 128.0]   # step 1 skipped on both ranks
    assert t0[0][1] == t0[1][1] != t0[2][1]              # parameters untouched by the skipped step, moved by the next
    assert s0 == s1 == 2                                  # the skipped step does not count (rolled back when settled)


# ------------------------------------------------------------------------------------------------- determinism ----
@pytest.mark.parametrize("backbone,amp", [("resnet18", False), ("resnet_tiny_bottleneck", True), ("vit_tiny_test", True),
                                          ("vit_tiny_test", False)])
def test_train_step_gradients_are_bit_reproducible(backbone, amp):
    """SURVEY §7 'deterministic summation order': the same step twice gives bit-identical gradients (weight gradients go
    through per-split slabs + an ordered second stage, statistics through per-tile partials; no float atomics on the path)."""
    cfg_model = dict(model=backbone, pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    torch.manual_seed(0)
    model = get_model(cfg_model, ["a", "b", "c"], DEV)
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    g = torch.Generator().manual_seed(9)
    hw = 96 if backbone == "resnet18" else 64
    x, y = torch.randn(16, 3, hw, hw, generator=g).to(DEV), torch.randint(0, 3, (16,), generator=g).to(DEV)
    model.train()
    grads = []
    for _ in range(3):
        for p in model.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            loss = crit(model(x), y)
        loss.backward()
        torch.cuda.synchronize()
        grads.append(model.arena.flat_grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    assert grads[0].abs().max().item() > 0


# ------------------------------------------------------------------------------------------------------- fp8 ----
def test_unicom_fp8_train_step_tracks_bf16_and_oracle():
    """BASELINE configs[4]'s numeric mode on a reduced unicom ViT whose Linear layers sit inside the fp8 GEMM envelope
    (dim 256): forward e4m3 x e4m3, data gradients e5m2 x e4m3, per-tensor scaling, bf16 everything else.  The fp8 step must
    stay close to the bf16 step (same kernels otherwise) and to the fp32 CPU oracle at the level fp8 operand rounding allows.
    The unicom architecture itself is restated from memory: parity of the ARCHITECTURE is unpinned; this test
    pins the arithmetic of the fp8 path against the oracle of that restatement."""
    from oracle.torch_models import OracleClassifier
    cfg_model = dict(model="unicom ViT-small-test", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    classes = ["a", "b", "c", "d"]
    torch.manual_seed(0)
    oracle = OracleClassifier(cfg_model, classes)
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(32, 3, 56, 56, generator=g), torch.randint(0, 4, (32,), generator=g)
    for net in (oracle,):
        for blk in net.emb_model.blocks:
            blk.drop_path.drop_prob = 0.0
    oracle.train()
    ref_out = oracle(x)
    torch.nn.functional.cross_entropy(ref_out, y).backward()
    ref_g = {n: p.grad.clone() for n, p in oracle.named_parameters()}
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    res = {}
    for mode in ("bf16", "fp8"):
        model = get_model(dict(cfg_model), classes, DEV)
        model.load_state_dict(oracle.state_dict())
        for blk in model.emb_model.blocks:
            blk.drop_path.drop_prob = 0.0
        model.train()
        model.fp8_linear = mode == "fp8"
        for _ in range(2):                          # second pass: delayed scaling in effect (scales from the first pass's amax)
            for p in model.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(x.to(DEV))
                loss = crit(out, y.to(DEV))
            loss.backward()
        torch.cuda.synchronize()
        eng = model._engines[torch.bfloat16]
        assert (len(eng._f8w) == 8) == (mode == "fp8")          # 2 blocks x (qkv, proj, fc1, fc2) took the fp8 kernel
        res[mode] = (out.detach().float().cpu(), loss.item(), {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()})

    def cos(a, b):
        num = sum((a[n].double() * b[n].double()).sum().item() for n in a)
        return num / (sum(a[n].double().pow(2).sum().item() for n in a) * sum(b[n].double().pow(2).sum().item() for n in b)) ** 0.5

    o16, l16, g16 = res["bf16"]
    o8, l8, g8 = res["fp8"]
    scale = ref_out.detach().abs().max().item()
    e16 = (o16 - ref_out.detach()).abs().max().item() / scale
    e8 = (o8 - ref_out.detach()).abs().max().item() / scale
    c16, c8 = cos(g16, ref_g), cos(g8, ref_g)
    print(f"\n[unicom small fp8] logits relerr bf16 {e16:.3e} fp8 {e8:.3e}; loss {l16:.4f} / {l8:.4f}; grad cosine vs oracle bf16 {c16:.5f} fp8 {c8:.5f}")
    assert e8 < 8e-2 and abs(l8 - l16) < 3e-2 * abs(l16)
    assert c8 > 0.97 and c16 > 0.995


@pytest.mark.parametrize("mode", ["bf16", "fp8"])
def test_unicom_branch_gradient_from_layernorm_backward_equals_separate_pass(mode, monkeypatch):
    """Stochastic depth active (rate 0.5) on the reduced unicom ViT (dim 256, two blocks): the branch gradient scale[b] * dx that
    LayerNorm backward writes next to dx — bf16 copy in the bf16 step (into the CONSUMER block's scratch set, behind the side-stream
    wait that guards it), e5m2 operand + bias column sums in the fp8 step — against the separate scale_rows / quantise + column-sum
    passes, same drop masks (same host seed): the bf16 step must give bit-identical gradients, the fp8 step bit-identical weight
    gradients (the operand bytes are the same) and bias gradients equal up to summation order."""
    from nkb_classification import hipnet
    cfg_model = dict(model="unicom ViT-small-test", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(32, 3, 56, 56, generator=g).to(DEV), torch.randint(0, 4, (32,), generator=g).to(DEV)
    grads = {}
    for fused in (True, False):
        monkeypatch.setattr(hipnet, "_LN_BWD_SCALED_COPY", fused)
        monkeypatch.setattr(hipnet, "_FP8_LN_BWD_QUANT", fused)
        torch.manual_seed(0)
        model = get_model(dict(cfg_model), ["a", "b", "c", "d"], DEV)
        for blk in model.emb_model.blocks:
            blk.drop_path.drop_prob = 0.5
        model.train()
        model.fp8_linear = mode == "fp8"
        torch.manual_seed(77)                       # the drop-path draws come from torch's host generator
        for _ in range(3):                          # fp8: the producers write the operands from the second step on
            for p in model.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = crit(model(x), y)
            loss.backward()
        torch.cuda.synchronize()
        eng = model._engines[torch.bfloat16]
        dropped = sum(int((eng.saved[f"b{i}.dp{j}"]["scale"] == 0).sum()) for i in range(2) for j in (1, 2))
        assert 0 < dropped < 4 * 32
        grads[fused] = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    for n in grads[True]:
        a, b = grads[True][n], grads[False][n]
        if mode == "bf16" or not n.endswith("bias"):
            assert torch.equal(a, b), n
        else:
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6 * max(1.0, b.abs().max().item()), msg=n)


def test_unicom_bf16_fused_epilogues_against_oracle_with_replayed_drop_masks():
    """The bf16 unicom step with every late-round-2 fusion in play on the reduced member (dim 256; the eight-phase core forced on
    for its small GEMMs): stochastic-depth scale in the proj / fc2 residual epilogue, ReLU6 in the fc1 epilogue and its mask in
    the fc2 data gradient, the scaled branch gradient from LayerNorm backward.  The engine's per-sample keep draws are replayed
    in the fp32 CPU oracle; logits within bf16 rounding of it, whole-model gradient cosine >= 0.999."""
    from oracle.torch_models import OracleClassifier
    cfg_model = dict(model="unicom ViT-small-test", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    classes = ["a", "b", "c", "d"]
    torch.manual_seed(0)
    oracle = OracleClassifier(cfg_model, classes)
    model = get_model(dict(cfg_model), classes, DEV)
    model.load_state_dict(oracle.state_dict())
    for net in (oracle, model):
        for blk in net.emb_model.blocks:
            blk.drop_path.drop_prob = 0.5
    g = torch.Generator().manual_seed(6)
    x, y = torch.randn(32, 3, 56, 56, generator=g), torch.randint(0, 4, (32,), generator=g)
    oracle.train(); model.train()
    hip.gemm8p_config(True, 1, 128)
    try:
        torch.manual_seed(31)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x.to(DEV))
            loss = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)(out, y.to(DEV))
        eng = model._engines[torch.bfloat16]
        dropped = 0
        for i, blk in enumerate(oracle.emb_model.blocks):
            s1, s2 = eng.saved[f"b{i}.dp1"]["scale"].cpu(), eng.saved[f"b{i}.dp2"]["scale"].cpu()
            dropped += int((s1 == 0).sum() + (s2 == 0).sum())
            keeps = iter([(s1 > 0).float(), (s2 > 0).float()])
            blk.drop_path.forward = (lambda mod, it: (lambda t: t * next(it).reshape(-1, 1, 1) / (1.0 - mod.drop_prob)))(blk.drop_path, keeps)
        assert 0 < dropped < 4 * 32
        ref = oracle(x)
        torch.nn.functional.cross_entropy(ref, y).backward()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        hip.gemm8p_config(True, 192, 768)
    scale = ref.detach().abs().max().item()
    err = (out.detach().float().cpu() - ref.detach()).abs().max().item() / scale
    ref_p = dict(oracle.named_parameters())
    num = den_a = den_b = 0.0
    for name, p in model.named_parameters():
        a, b = p.grad.detach().cpu().double(), ref_p[name].grad.double()
        num += (a * b).sum().item(); den_a += a.pow(2).sum().item(); den_b += b.pow(2).sum().item()
    cos = num / (den_a * den_b) ** 0.5
    print(f"\n[unicom small bf16, fused epilogues] logits relerr {err:.3e}, gradient cosine {cos:.6f}, dropped {dropped}")
    assert err < 3e-2 and cos > 0.999


def test_timm_vit_bf16_gelu_epilogue_against_oracle():
    """timm-layout ViT (dim 256, GELU MLP) in bf16 with the eight-phase core forced on for its small GEMMs: fc1 then writes gelu(pre)
    and gelu'(pre) from its epilogue (Abramowitz-Stegun erf) and the fc2 data gradient multiplies by the saved derivative — logits and
    whole-model gradients against the fp32 CPU oracle (exact erf)."""
    from oracle.torch_models import OracleClassifier
    cfg_model = dict(model="vit_small_test", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    classes = ["a", "b", "c", "d"]
    torch.manual_seed(0)
    oracle = OracleClassifier(cfg_model, classes)
    model = get_model(dict(cfg_model), classes, DEV)
    model.load_state_dict(oracle.state_dict())
    g = torch.Generator().manual_seed(8)
    x, y = torch.randn(64, 3, 64, 64, generator=g), torch.randint(0, 4, (64,), generator=g)
    oracle.train(); model.train()
    ref = oracle(x)
    torch.nn.functional.cross_entropy(ref, y).backward()
    hip.gemm8p_config(True, 1, 128)
    try:
        assert hip.linear_gelu_fused_ok(hip.BF16, 64 * 17, 256, 1024)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x.to(DEV))
            loss = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)(out, y.to(DEV))
        eng = model._engines[torch.bfloat16]
        assert "gp" in eng.saved["b0.act"] and "x" not in eng.saved["b0.act"]      # the epilogue form ran: no pre-activation kept
        loss.backward()
        torch.cuda.synchronize()
    finally:
        hip.gemm8p_config(True, 192, 768)
    scale = ref.detach().abs().max().item()
    err = (out.detach().float().cpu() - ref.detach()).abs().max().item() / scale
    ref_p = dict(oracle.named_parameters())
    num = den_a = den_b = 0.0
    for name, p in model.named_parameters():
        a, b = p.grad.detach().cpu().double(), ref_p[name].grad.double()
        num += (a * b).sum().item(); den_a += a.pow(2).sum().item(); den_b += b.pow(2).sum().item()
    cos = num / (den_a * den_b) ** 0.5
    print(f"\n[timm ViT small bf16, GELU epilogue] logits relerr {err:.3e}, gradient cosine {cos:.6f}")
    assert err < 3e-2 and cos > 0.999


def test_timm_vit_fp8_train_step_tracks_bf16_and_oracle():
    """The fp8 mode on a reduced timm-layout ViT (dim 256, class token, GELU MLP, qkv WITH bias): 17 tokens x 128 images = 2176 rows
    inside the fp8 GEMM envelope.  Exercises what the unicom member does not: the qkv bias gradient from the attention backward
    kernel's stores, LayerNorm backward's fp8 operand without a stochastic-depth scale, the GELU pass next to an fp8 fc1.  Same
    bounds as the unicom test: fp8 close to bf16 (same kernels otherwise) and to the fp32 CPU oracle at the fp8 rounding level."""
    from oracle.torch_models import OracleClassifier
    cfg_model = dict(model="vit_small_test", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    classes = ["a", "b", "c", "d"]
    torch.manual_seed(0)
    oracle = OracleClassifier(cfg_model, classes)
    g = torch.Generator().manual_seed(5)
    x, y = torch.randn(128, 3, 64, 64, generator=g), torch.randint(0, 4, (128,), generator=g)
    oracle.train()
    ref_out = oracle(x)
    torch.nn.functional.cross_entropy(ref_out, y).backward()
    ref_g = {n: p.grad.clone() for n, p in oracle.named_parameters()}
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    res = {}
    for mode in ("bf16", "fp8"):
        model = get_model(dict(cfg_model), classes, DEV)
        model.load_state_dict(oracle.state_dict())
        model.train()
        model.fp8_linear = mode == "fp8"
        for _ in range(2):                          # second pass: delayed scaling in effect, operands written by their producers
            for p in model.parameters():
                p.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(x.to(DEV))
                loss = crit(out, y.to(DEV))
            loss.backward()
        torch.cuda.synchronize()
        eng = model._engines[torch.bfloat16]
        assert (len(eng._f8w) == 8) == (mode == "fp8")          # 2 blocks x (qkv, proj, fc1, fc2) took the fp8 kernel
        res[mode] = (out.detach().float().cpu(), loss.item(), {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()})

    def cos(a, b):
        num = sum((a[n].double() * b[n].double()).sum().item() for n in a)
        return num / (sum(a[n].double().pow(2).sum().item() for n in a) * sum(b[n].double().pow(2).sum().item() for n in b)) ** 0.5

    o16, l16, g16 = res["bf16"]
    o8, l8, g8 = res["fp8"]
    scale = ref_out.detach().abs().max().item()
    e16 = (o16 - ref_out.detach()).abs().max().item() / scale
    e8 = (o8 - ref_out.detach()).abs().max().item() / scale
    c16, c8 = cos(g16, ref_g), cos(g8, ref_g)
    # the qkv bias gradients (attention backward's column sums in fp8 mode) against the oracle's, tensor by tensor
    qb = [n for n in ref_g if n.endswith("attn.qkv.bias")]
    cq = min(torch.nn.functional.cosine_similarity(g8[n].flatten().double(), ref_g[n].flatten().double(), dim=0).item() for n in qb)
    print(f"\n[timm ViT small fp8] logits relerr bf16 {e16:.3e} fp8 {e8:.3e}; loss {l16:.4f} / {l8:.4f}; grad cosine vs oracle bf16 {c16:.5f} "
          f"fp8 {c8:.5f}; qkv bias {cq:.5f}")
    assert len(qb) == 2 and cq > 0.97
    # measured: logits 7.1e-3 (bf16) / 9.1e-2 (fp8) of the largest logit, loss 1.8944 / 1.8727, gradient cosine 0.99998 / 0.99769,
    # qkv bias 0.998.  The logit bound is looser than the unicom member's (2.2e-2 there): 17-token rows, K = 256 contractions and one
    # per-tensor scale shared by the class token and the patch tokens leave e4m3's 2^-4 relative step less room to average out.
    assert e8 < 0.15 and e16 < 2e-2 and abs(l8 - l16) < 3e-2 * abs(l16)
    assert c8 > 0.97 and c16 > 0.995


@pytest.mark.parametrize("mode", ["bf16", "fp8"])
def test_unicom_vit_l14_full_size_train_steps(mode):
    """BASELINE configs[4]'s model at full size (unicom ViT-L/14: 24 blocks x 1024, 256 tokens, 572 M parameters; one GPU, a small
    batch): three optimizer steps on one batch must run through every full-size kernel path (the 262 144-deep split-K feature
    head, 16-head attention at T = 256, the eight-phase GEMMs, fp8 quantisation in fp8 mode), keep every value finite and
    drive the loss down.  Architecture parity is unpinned (the unicom package is absent; restated from memory and
    pinned by parameter count / key names in tests/test_oracle_golden.py) — this is the execution check of that restatement."""
    cfg_model = dict(model="unicom ViT-L/14", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    torch.manual_seed(0)
    model = get_model(cfg_model, [str(i) for i in range(10)], DEV)
    assert sum(p.numel() for p in model.emb_model.parameters()) == 572_328_448
    for blk in model.emb_model.blocks:
        blk.drop_path.drop_prob = 0.0
    opt = get_optimizer(model, dict(type="sgd", lr=0.05))
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(16, 3, 224, 224, generator=g).to(DEV), torch.randint(0, 10, (16,), generator=g).to(DEV)
    model.train()
    model.fp8_linear = mode == "fp8"
    losses = []
    for _ in range(4):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(model(x), y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    assert all(math.isfinite(v) for v in losses) and torch.isfinite(model.arena.flat_param).all().item()
    assert losses[-1] < losses[0], losses
    eng = model._engines[torch.bfloat16]
    assert len(eng._f8w) == (96 if mode == "fp8" else 0)


def test_bf16_gradient_exchange_kernels_on_the_gpu_single_rank_rccl():
    """The GPU form of parallel.GradReducer's bf16 bucket exchange (cast by nkb_wprep, all-to-all, fp32 sum of the received slices by
    nkb_bucket_sum_bf16, all-gather, widening back into the arena) through a real one-rank RCCL group — the CPU / gloo tests cover
    the protocol, not these kernels: (a) nkb_bucket_sum_bf16 against torch on W = 4 slices, fp32 and bf16 outputs, ragged length;
    (b) a whole exchange on an arena slice that starts 64-aligned and ends at an odd element: with one rank the result is the
    gradient rounded to bf16 once."""
    import torch.distributed as dist
    from nkb_classification.parallel import GradReducer
    g = torch.Generator().manual_seed(3)
    W, chunk, n = 4, 1016, 1003                                  # stride a multiple of 8, length not
    parts = torch.randn(W * chunk, generator=g).to(DEV).to(torch.bfloat16)
    out32, out16 = torch.full((n,), 7.0, device=DEV), torch.zeros(chunk, device=DEV, dtype=torch.bfloat16)
    hip.bucket_sum_bf16(parts, chunk, W, out32, out16, n)
    ref = torch.zeros(n, device=DEV)
    for p_ in range(W):                                          # fp32 accumulation in part order
        ref += parts[p_ * chunk:p_ * chunk + n].float()
    torch.cuda.synchronize()
    assert torch.equal(out32, ref) and torch.equal(out16[:n], ref.to(torch.bfloat16))
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    dist.init_process_group("nccl", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{31000 + os.getpid() % 2000}",
                            device_id=torch.device(DEV))
    try:
        total = 64 * 300 + 10                                    # ends in a 10-element classifier bias
        flat = torch.randn(total, generator=g).to(DEV) * 3.0
        model = types.SimpleNamespace(arena=types.SimpleNamespace(flat_grad=flat, total=total), grad_ready_hook=None)
        red = GradReducer(model, None, bucket_bytes=4 * 4096, bucket_dtype="bf16")
        assert red.bf16_buckets and red.world == 1
        want = flat.clone()
        lo = 64 * 37
        want[lo:] = want[lo:].to(torch.bfloat16).float()
        red._exchange_bf16(flat[lo:])                            # (world 1: the hooks return early, so call the exchange itself)
        torch.cuda.synchronize()
        assert torch.equal(flat, want)
        # (c) the exchange inside a train step, recorded into the backward launch plan (ADVICE r3 / VERDICT r3 #6): the reducer is
        # attached to a real model with small bf16 buckets and told it has two ranks (bench.py's NKB_FORCE_REDUCER rehearsal), so
        # the bucket hooks fire during backward.  From the third step on the staging buffers are the ones the first steps made (no
        # allocation inside _exchange_bf16), the backward plan IS kept with the hooks in it, and — one rank — the parameters follow
        # the run without a reducer up to the one bf16 rounding of each gradient.
        cfg_model = dict(model="resnet_tiny_bottleneck", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                         classifier_initialization="kaiming_normal_", task="single")
        crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
        gx = torch.Generator().manual_seed(5)
        x, y = torch.randn(8, 3, 64, 64, generator=gx).to(DEV), torch.randint(0, 3, (8,), generator=gx).to(DEV)

        def run(with_reducer):
            torch.manual_seed(0)
            model = get_model(dict(cfg_model), ["a", "b", "c"], DEV)
            opt = get_optimizer(model, dict(type="sgd", lr=0.05))
            red2, inside = None, []
            if with_reducer:
                red2 = GradReducer(model, opt, bucket_bytes=64 << 10, bucket_dtype="bf16")
                red2.world = 2
                opt.grad_scale = 1.0
                inner = red2._exchange_bf16

                def counted(gs):
                    before = torch.cuda.memory_stats(DEV).get("allocation.all.allocated", 0)
                    inner(gs)
                    inside.append(torch.cuda.memory_stats(DEV).get("allocation.all.allocated", 0) - before)
                red2._exchange_bf16 = counted
            model.train()
            per_step = []
            for _ in range(6):
                n0 = len(inside)
                opt.zero_grad()
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = crit(model(x), y)
                loss.backward()
                opt.step()
                per_step.append(inside[n0:])
            torch.cuda.synchronize()
            return model, red2, per_step

        m1, red2, per_step = run(True)
        eng = m1._engines[torch.bfloat16]
        assert all(len(v) >= 2 for v in per_step), per_step     # several buckets went out in every step
        assert all(a == 0 for v in per_step[2:] for a in v), per_step      # ... and none of them allocated after step 2
        assert red2.stage_allocs == len(red2._stage) <= 8
        assert any(k[0] == "bwd" for k in eng.plans), list(eng.plans)       # the backward plan is kept with the bucket hooks in it
        m0, _, _ = run(False)
        assert torch.isfinite(m1.arena.flat_param).all().item()
        # (each gradient was rounded to bf16 once, six SGD steps at lr 0.05 on a step that is itself bf16: measured 3e-3 worst element)
        torch.testing.assert_close(m1.arena.flat_param, m0.arena.flat_param, rtol=0, atol=1e-2)
        assert ((m1.arena.flat_param - m0.arena.flat_param).norm() / m0.arena.flat_param.norm()).item() < 1e-3
    finally:
        dist.destroy_process_group()


def test_host_op_library_calls_are_not_recorded_twice():
    """A Python operation recorded into a launch plan (hip.host_op) is re-run at every replay; library calls made INSIDE it while
    the plan is being recorded (the bf16 gradient exchange casts and sums through libnkbhip on temporaries) must not also become
    table entries — replayed, those would run again on pointers freed when the operation returned."""
    a = torch.arange(12, device=DEV, dtype=torch.float32).reshape(3, 4).contiguous()
    b = torch.zeros(3, 4, device=DEV)
    calls = []

    def op():
        tmp = a.clone()                                  # a temporary that dies with the call
        hip.add2d(tmp, b, 3, 4, 4, 4)                    # b += tmp through the library
        calls.append(1)

    hip.record_begin()
    try:
        hip.host_op(op)
        hip.add2d(a, b, 3, 4, 4, 4)                      # a direct call: this one IS a table entry
    finally:
        plan = hip.record_end({})
    kinds = [seg[0] for seg in plan.segments]
    assert kinds.count(2) == 1 and sum(seg[2] for seg in plan.segments if seg[0] == 0) == 1, kinds
    hip.replay(plan, {})
    torch.cuda.synchronize()
    assert len(calls) == 2 and torch.equal(b, 4 * a)     # recorded once + replayed once, each time: op (+a) and the entry (+a)


# ------------------------------------------------------------------------------------------------- launch plans ----
@pytest.mark.parametrize("backbone", ["resnet_tiny_bottleneck", "vit_tiny_test"])
def test_launch_plans_replay_exactly_across_shape_and_mode_changes(backbone, monkeypatch):
    """Recorded launch plans (hip.Plan) vs the Python path on the same sequence: full batches (recorded after two eager
    steps, then replayed), a partial last batch (another shape: its own eager steps, workspace reallocation invalidates the
    plans), an evaluation pass in between, a frozen-backbone step, classifier dropout (seeds drawn per replay) — the
    parameters after the whole sequence must be bit-identical with plans on and off."""
    from nkb_classification import model as model_mod
    cfg_model = dict(model=backbone, pretrained=False, backbone_dropout=0.0, classifier_dropout=0.3,
                     classifier_initialization="kaiming_normal_", task="single")
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    g = torch.Generator().manual_seed(12)
    xs = [torch.randn(8, 3, 64, 64, generator=g).to(DEV) for _ in range(3)] + [torch.randn(5, 3, 64, 64, generator=g).to(DEV)]
    ys = [torch.randint(0, 3, (x.shape[0],), generator=g).to(DEV) for x in xs]
    schedule = [0, 1, 2, 0, 1, 3, "eval", 2, 0, "freeze", 1, 2, 0, 1, "unfreeze", 2, 0, 1, 3, 3, 2]

    def run(plans_on, c_replay=True):
        monkeypatch.setattr(model_mod, "_PLANS", plans_on)
        monkeypatch.setattr(hip, "_PLAN_C", c_replay)        # nkb_plan_run (csrc/plan.hip) vs the per-entry ctypes loop
        torch.manual_seed(0)
        model = get_model(dict(cfg_model), ["a", "b", "c"], DEV)
        opt = get_optimizer(model, dict(type="nadam", lr=1e-3, weight_decay=0.01))
        torch.manual_seed(77)                        # dropout seed stream
        model.train()
        used = 0
        for item in schedule:
            if item == "eval":
                model.eval()
                with torch.no_grad():
                    model(xs[0])
                model.train()
                continue
            if item in ("freeze", "unfreeze"):
                model.set_backbone_state(item)
                continue
            opt.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = crit(model(xs[item]), ys[item])
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        eng = model._engines[torch.bfloat16]
        kinds = [seg[0] for ent in eng.plans.values() for seg in ent[0].segments]
        return model.arena.flat_param.clone(), len(eng.plans), kinds

    p_on, n_on, k_on = run(True)
    p_py, n_py, k_py = run(True, c_replay=False)
    p_off, n_off, _ = run(False)
    assert n_off == 0 and n_on >= 2 and n_py == n_on # at least one forward and one backward plan were recorded and replayed
    assert k_on.count(0) >= 2 and k_on.count(1) == 0 # C replay: every recorded call sits in a table segment (none left to ctypes)
    assert k_py.count(0) == 0 and k_py.count(1) > 50 # Python replay: one ctypes call per entry
    assert torch.equal(p_on, p_off) and torch.equal(p_py, p_off)
