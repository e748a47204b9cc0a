"""End-to-end parity of the HIP train step (through nkb_classification's drop-in API) against the CPU oracle
and the golden trajectories captured from the reference engine (tests/golden/g4_engine.json).

Bar (BASELINE north_star): logits within 1e-3 relative in fp32, argmax bit-exact.
"""
import math
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from nkb_classification.engine import train_epoch, val_epoch  # noqa: E402
from nkb_classification.logging import BaseLogger  # noqa: E402
from nkb_classification.losses import get_loss  # noqa: E402
from nkb_classification.model import get_model  # noqa: E402
from nkb_classification.utils import get_optimizer, get_scheduler  # noqa: E402
from oracle.torch_engine import synthetic_batches  # noqa: E402
from oracle.torch_models import OracleClassifier  # noqa: E402

DEV = "cuda:0"


def _cfg(task, amp=False, log_gradients=False):
    return types.SimpleNamespace(task=task, enable_mixed_presicion=amp, log_gradients=log_gradients,
                                 show_full_current_loss_in_terminal=False)


def _pair(cfg_model, classes, seed=0):
    torch.manual_seed(seed)
    oracle = OracleClassifier(cfg_model, classes)
    model = get_model(dict(cfg_model), classes, DEV)
    model.load_state_dict(oracle.state_dict())
    return oracle, model


def _relerr(a, b):
    a, b = torch.as_tensor(a, dtype=torch.float64), torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


@pytest.mark.parametrize("backbone", ["resnet_tiny_basic", "resnet_tiny_bottleneck", "resnet18", "vit_tiny_test",
                                      "unicom ViT-tiny-test",     # model.py:77-79 family (stochastic depth switched off here)
                                      # odd sizes: clipped pooling windows, odd parity-class grids of the stride-2 data
                                      # gradients, ceil-sized sub-grid shortcut gradients, the packed stem's pad column
                                      "resnet_tiny_basic@54x3", "resnet_tiny_bottleneck@70x3", "resnet_tiny_bottleneck@45x2"])
def test_single_step_gradients_match_oracle_fp32(backbone):
    """Truth = the oracle evaluated in float64.  ReLU/max-pool decisions on near-zero pre-activations make the
    problem mildly ill-conditioned, so the HIP fp32 path is held to the same distance from the float64 truth as
    torch's own CPU fp32 path instead of to a fixed distance from the fp32 CPU result (see the bounds below).
    Logits keep the 1e-3 / exact-argmax bar."""
    shape = None
    if "@" in backbone:
        backbone, shape = backbone.split("@")
        shape = tuple(int(v) for v in shape.split("x"))
    cfg_model = dict(model=backbone, pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    classes = ["a", "b", "c"]
    o32, model = _pair(cfg_model, classes)
    # zero_init_last makes the residual branches vanish at init; randomise BN affine params so every path carries signal
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in o32.parameters():
            if p.dim() == 1:
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.5)
    model.load_state_dict(o32.state_dict())
    o64 = OracleClassifier(cfg_model, classes).double()
    o64.load_state_dict(o32.state_dict())
    hw = 64 if backbone != "resnet18" else 96
    nb = 4
    if backbone.startswith("unicom"):
        hw = 56                                    # 16 tokens of 14x14 pixels; pos_embed fixes the input size
        for net in (o32, o64, model):
            for blk in net.emb_model.blocks:
                blk.drop_path.drop_prob = 0.0      # the draw is replayed in test_unicom_drop_path_matches_oracle
    if shape is not None:
        hw, nb = shape
    x = torch.randn(nb, 3, hw, hw + (3 if shape is not None else 0), generator=g)      # odd cases are non-square too
    y = torch.randint(0, 3, (nb,), generator=g)
    o32.train(); o64.train(); model.train()
    ref32 = o32(x)
    torch.nn.functional.cross_entropy(ref32, y).backward()
    ref64 = o64(x.double())
    torch.nn.functional.cross_entropy(ref64, y).backward()
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    out = model(x.to(DEV))
    loss = crit(out, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert _relerr(out.detach().cpu(), ref32.detach()) < 1e-3
    assert out.argmax(-1).cpu().tolist() == ref32.argmax(-1).tolist()
    p64, p32 = dict(o64.named_parameters()), dict(o32.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in p64.values())
    num = den = 0.0
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        assert p.grad.shape == p32[name].grad.shape
        ref = p64[name].grad
        scale = max(ref.abs().max().item(), 1e-6 * gmax)   # exact invariances give |grad| ~ 1e-16: compare to the model's scale
        e_hip = (p.grad.cpu().double() - ref).abs().max().item() / scale
        e_cpu = (p32[name].grad.double() - ref).abs().max().item() / scale
        # a single flipped ReLU decision (observed: 1 element in 65k, either implementation) moves a small layer's
        # gradient by ~1/n_terms, so the per-tensor bound only screens for O(1) errors (wrong indexing, lost terms) ...
        assert e_hip <= max(2e-2, 4 * e_cpu), (name, e_hip, e_cpu)
        num += (p.grad.cpu().double() - ref).pow(2).sum().item()
        den += ref.pow(2).sum().item()
    # ... and the whole-model gradient is held to 3e-3 in the L2 sense
    assert (num / den) ** 0.5 < 3e-3, (num / den) ** 0.5
    # running statistics advanced identically
    ob, mb = dict(o32.named_buffers()), dict(model.named_buffers())
    for k in ob:
        torch.testing.assert_close(mb[k].cpu().to(ob[k].dtype), ob[k], rtol=1e-4, atol=1e-5, msg=k)


def _run_case(case, amp=False):
    classes = case["classes"]
    task = "multi" if isinstance(classes, dict) else "single"
    oracle, model = _pair(case["cfg_model"], classes, seed=case["seed"])
    n_cls = {t: len(c) for t, c in classes.items()} if task == "multi" else len(classes)
    train = synthetic_batches(case["n_images"], case["batch"], n_cls, seed=1234, hw=case["hw"])
    val = synthetic_batches(2 * case["batch"], case["batch"], n_cls, seed=4321, hw=case["hw"])
    opt = get_optimizer(model, case["opt_cfg"])
    sch = get_scheduler(opt, dict(type="cosine", n_epochs=case["n_epochs_cos"]))
    crit = get_loss(case["crit_cfg"], DEV)
    cfg = _cfg(task, amp=amp, log_gradients=case["epochs"][0].get("grad_total") is not None)
    logger = BaseLogger(cfg, classes)
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    out = []
    for _ in case["epochs"]:
        tr = train_epoch(model, train, opt, sch, scaler, crit, DEV, cfg, logger)
        lr_after = [g["lr"] for g in opt.param_groups]
        va = val_epoch(model, val, crit, DEV, cfg, logger)
        out.append((tr, va, lr_after))
    model.eval()
    with torch.no_grad():
        logits = model(val[0][0].to(DEV))
    return out, logits, model


def _check_case(case, tol):
    out, logits, model = _run_case(case)
    multi = isinstance(case["classes"], dict)
    for (tr, va, lr_after), gold in zip(out, case["epochs"]):
        if multi:
            for k in gold["train_running_loss"]:
                assert _relerr(tr["running_loss"][k], gold["train_running_loss"][k]) < tol, k
                assert _relerr(va["running_loss"][k], gold["val_running_loss"][k]) < tol, k
            for k in gold["train_ground_truth"]:
                assert tr["ground_truth"][k] == gold["train_ground_truth"][k]
                assert _relerr(va["confidences"][k], gold["val_confidences"][k]) < tol
        else:
            assert _relerr(tr["running_loss"], gold["train_running_loss"]) < tol
            assert _relerr(va["running_loss"], gold["val_running_loss"]) < tol
            assert tr["ground_truth"] == gold["train_ground_truth"]
            assert _relerr(va["confidences"], gold["val_confidences"]) < tol
        assert lr_after == pytest.approx(gold["lr_after"], rel=1e-12)
        if "grad_total" in gold:
            got = [float(v) for v in tr["metrics_grad_log"]["Gradients/Total"]]
            assert _relerr(got, gold["grad_total"]) < 5 * tol
    if multi:
        for k, v in case["final_val_logits"].items():
            assert _relerr(logits[k].cpu(), v) < tol
            assert logits[k].argmax(-1).cpu().tolist() == case["final_val_argmax"][k]
    else:
        assert _relerr(logits.cpu(), case["final_val_logits"]) < tol
        assert logits.argmax(-1).cpu().tolist() == case["final_val_argmax"]
    sd = model.state_dict()
    for k, v in case["param_norms"].items():
        assert abs(float(sd[k].float().norm()) - v) <= tol * max(1.0, abs(v)), k


def test_golden_trajectory_tiny_basic_single(golden):
    _check_case(golden("g4_engine")["tiny_basic_single"], 1e-3)


def test_golden_trajectory_tiny_bottleneck_multi(golden):
    _check_case(golden("g4_engine")["tiny_bottleneck_multi"], 1e-3)


def test_golden_trajectory_tiny_vit_single(golden):
    _check_case(golden("g4_engine")["tiny_vit_single"], 1e-3)


def test_vit_base_forward_matches_oracle():
    """Full-size timm-layout ViT-B/16 (197 tokens, 12 heads): eval-mode logits vs the CPU oracle, fp32 and bf16."""
    cfg_model = dict(model="vit_base_patch16_224", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    oracle, model = _pair(cfg_model, [str(i) for i in range(10)])
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    oracle.eval(); model.eval()
    with torch.no_grad():
        ref = oracle(x)
        out = model(x.to(DEV))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out16 = model(x.to(DEV))
    assert _relerr(out.cpu(), ref) < 1e-3 and out.argmax(-1).cpu().tolist() == ref.argmax(-1).tolist()
    assert _relerr(out16.cpu(), ref) < 5e-2


def test_golden_trajectory_config1_resnet18(golden):
    """BASELINE config 1: ResNet-18, 2 classes, 64 synthetic 224x224 images, bs=8, fp32, NAdam."""
    _check_case(golden("g4_engine")["config1_resnet18"], 1e-3)


def test_bf16_mode_tracks_fp32(golden):
    case = golden("g4_engine")["tiny_basic_single"]
    out, logits, _ = _run_case(case, amp=True)
    gold = case["epochs"][0]
    got = out[0][0]["running_loss"]
    assert all(math.isfinite(v) for v in got)
    assert _relerr(got, gold["train_running_loss"]) < 0.1


def test_frozen_backbone_only_updates_head():
    cfg_model = dict(model="resnet_tiny_bottleneck", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    oracle, model = _pair(cfg_model, ["a", "b"])
    model.set_backbone_state("freeze"); oracle.set_backbone_state("freeze")
    g = torch.Generator().manual_seed(9)
    x, y = torch.randn(4, 3, 64, 64, generator=g), torch.randint(0, 2, (4,), generator=g)
    model.train(); oracle.train()
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    crit(model(x.to(DEV)), y.to(DEV)).backward()
    torch.nn.functional.cross_entropy(oracle(x), y).backward()
    assert all(p.grad is None for p in model.emb_model.parameters())
    for (n, p), (_, q) in zip(model.classifier.named_parameters(), oracle.classifier.named_parameters()):
        assert _relerr(p.grad.cpu(), q.grad) < 1e-3, n
    # BN running stats still advance while frozen (model.py:59-64 only flips requires_grad)
    torch.testing.assert_close(model.emb_model.bn1.running_mean.cpu(), oracle.emb_model.bn1.running_mean, rtol=1e-4, atol=1e-5)


def test_degenerate_batches_fail_loudly_or_run():
    """An empty batch or a non-RGB input is a clear error (no zero-sized kernel launch); a single-image batch (the last
    partial batch of an epoch) trains — BatchNorm over one image and all."""
    cfg_model = dict(model="resnet_tiny_bottleneck", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    model = get_model(cfg_model, ["a", "b", "c"], DEV).train()
    with pytest.raises(RuntimeError, match="non-empty batch"):
        model(torch.zeros(0, 3, 64, 64, device=DEV))
    with pytest.raises(RuntimeError, match="3-channel"):
        model(torch.zeros(2, 1, 64, 64, device=DEV))
    out = model(torch.randn(1, 3, 64, 64, device=DEV))
    out.sum().backward()
    torch.cuda.synchronize()
    assert out.shape == (1, 3) and torch.isfinite(out).all()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_cpu_forward_fails_loudly():
    cfg_model = dict(model="resnet_tiny_basic", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    model = get_model(cfg_model, ["a", "b"], "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.randn(1, 3, 64, 64))


def test_train_py_end_to_end(tmp_path):
    """The config-driven entry point (train.py -cfg ...) runs two epochs and writes the checkpoint files."""
    import subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1] / "nkb-classification_amd"
    cfg = (root / "configs" / "synthetic_singletask_config.py").read_text().replace(
        '"runs/synthetic_single"', repr(str(tmp_path / "exp")))
    (tmp_path / "cfg_e2e.py").write_text(cfg)
    r = subprocess.run([sys.executable, str(root / "train.py"), "-cfg", str(tmp_path / "cfg_e2e.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    exp = tmp_path / "exp"
    assert (exp / "weights" / "last.pth").exists() and (exp / "classes.json").exists()
    assert (exp / "weights" / "scripted_last.pt").exists()          # train.py:66-73 TorchScript archive
    lines = (exp / "metrics.csv").read_text().strip().splitlines()
    assert len(lines) == 3 and "\t" in lines[0]
    sd = torch.load(exp / "weights" / "last.pth", map_location="cpu")
    assert "emb_model.layer4.1.bn2.running_var" in sd and "classifier.1.weight" in sd
    # the TorchScript archive is the plain-torch twin of the HIP model that wrote it (scripted.py): its logits on a fixed batch
    # equal those of the HIP model rebuilt from last.pth (fp32, eval mode) to 1e-3 — what configs/eval_config.py:87-90 /
    # model.py:163-164 ("scripted": path) would load in the reference
    scripted = torch.jit.load(str(exp / "weights" / "scripted_last.pt"), map_location="cpu").eval()
    hip_model = get_model(dict(task="single", model="resnet18", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                               classifier_initialization="kaiming_normal_", checkpoint=str(exp / "weights" / "last.pth")),
                          [str(i) for i in range(10)], DEV).eval()
    xb = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        ref_logits = scripted(xb)
        hip_logits = hip_model(xb.to(DEV)).float().cpu()
    assert ref_logits.shape == hip_logits.shape == (6, 10)
    rel = ((hip_logits - ref_logits).abs().max() / ref_logits.abs().max()).item()
    assert rel < 1e-3, rel
    assert hip_logits.argmax(-1).tolist() == ref_logits.argmax(-1).tolist()
    # eval.py -cfg (the reference's eval.py:27-52): the checkpoint just written, one validation epoch, metrics.json
    import json
    ecfg = cfg.replace('"classifier_initialization": "kaiming_normal_"}',
                       '"classifier_initialization": "kaiming_normal_", "checkpoint": %r}' % str(exp / "weights" / "last.pth"))
    ecfg += "\nsave_path = %r\n" % str(tmp_path / "eval_out")
    (tmp_path / "cfg_eval.py").write_text(ecfg)
    r = subprocess.run([sys.executable, str(root / "eval.py"), "-cfg", str(tmp_path / "cfg_eval.py")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    m = json.loads((tmp_path / "eval_out" / "metrics.json").read_text())
    assert "epoch_acc" in m and 0.0 <= m["epoch_acc"] <= 1.0
    last = dict(zip(lines[0].split("\t"), lines[-1].split("\t")))          # the same validation set, the same weights
    key = next((k for k in last if "val" in k.lower() and "acc" in k.lower() and "balanced" not in k.lower()), None)
    if key is not None:
        assert abs(float(last[key]) - m["epoch_acc"]) < 1e-6, (key, last[key], m["epoch_acc"])


def _ddp_worker(rank, world, port, q, backbone="resnet_tiny_bottleneck"):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # gloo moves CUDA tensors through the host
    try:
        from nkb_classification.parallel import GradReducer
        cfg_model = dict(model=backbone, pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                         classifier_initialization="kaiming_normal_", task="single")
        torch.manual_seed(0)
        model = get_model(cfg_model, ["a", "b", "c"], DEV)
        hw = 56 if backbone.startswith("unicom") else 64
        if backbone.startswith("unicom"):
            for blk in model.emb_model.blocks:
                blk.drop_path.drop_prob = 0.0       # the two backward passes below must see the same function
        with torch.no_grad():                       # non-trivial BN affine parameters so every path carries signal
            g0 = torch.Generator().manual_seed(3)
            for p in model.parameters():
                if p.dim() == 1:
                    p.copy_((torch.rand(p.shape, generator=g0) * 0.5 + 0.5).to(p.device))
        crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
        opt = get_optimizer(model, dict(type="sgd", lr=0.1))
        g = torch.Generator().manual_seed(100 + rank)
        x, y = torch.randn(4, 3, hw, hw, generator=g).to(DEV), torch.randint(0, 3, (4,), generator=g).to(DEV)
        model.train()
        # local gradients first (no reducer attached)
        crit(model(x), y).backward()
        torch.cuda.synchronize()
        local = model.arena.flat_grad.clone()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        expect_sum = sum(gathered)
        params_before = model.arena.flat_param.clone()
        # same step through the reducer: ranges are all-reduced while backward runs, the optimizer applies 1/world
        opt.zero_grad()
        red = GradReducer(model, opt)
        crit(model(x), y).backward()
        opt.step()
        torch.cuda.synchronize()
        got_sum = model.arena.flat_grad
        ok_grad = torch.allclose(got_sum, expect_sum, rtol=1e-4, atol=1e-6)
        ok_step = torch.allclose(model.arena.flat_param, params_before - 0.1 * expect_sum / world, rtol=1e-4, atol=1e-6)
        q.put((rank, bool(ok_grad), bool(ok_step), opt.grad_scale))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backbone", ["resnet_tiny_bottleneck", "vit_tiny_test", "unicom ViT-tiny-test"])
def test_data_parallel_two_ranks_on_one_gpu(backbone):
    """world_size 2 with both ranks on cuda:0 (gloo transport): the HIP model's backward hooks + GradReducer + fused
    optimizer reproduce 'average of the per-rank gradients, then SGD' exactly — for every backbone family's
    gradient-ready notification order."""
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, backbone)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True, 0.5), (1, True, True, 0.5)]


def test_vit_backbone_dropout_matches_oracle_with_replayed_masks():
    """backbone_dropout > 0 on a ViT (the reference's sample configs use 0.1; set_dropout rewrites every nn.Dropout of
    timm's VisionTransformer): run the HIP model in train mode, then replay its keep masks inside the CPU oracle at the
    same six kinds of sites — logits and every parameter gradient must agree (fp32)."""
    cfg_model = dict(model="vit_tiny_test", pretrained=False, backbone_dropout=0.25, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    classes = ["a", "b", "c"]
    oracle, model = _pair(cfg_model, classes, seed=3)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(4, 3, 64, 64, generator=g)
    y = torch.randint(0, 3, (4,), generator=g)
    model.train(); oracle.train()
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    out = model(x.to(DEV))
    crit(out, y.to(DEV)).backward()
    torch.cuda.synchronize()
    saved = model._active.saved
    p = 0.25

    class Replay(torch.nn.Module):
        def __init__(self, mask):
            super().__init__()
            self.mask = mask.float().cpu()

        def forward(self, t):
            m = self.mask
            if m.dim() == 3 and t.dim() == 4:                    # attention: [B*H, T, Tp] -> [B, H, T, T]
                m = m[:, :, : t.shape[-1]].reshape(t.shape)
            return t * m.reshape(t.shape) / (1 - p)

    vit = oracle.emb_model
    keys = ["pos_drop", "head_drop"]
    vit.pos_drop = Replay(saved["pos_drop"]["mask"])
    vit.head_drop = Replay(saved["head_drop"]["mask"])
    for i, blk in enumerate(vit.blocks):
        blk.attn.attn_drop = Replay(saved[f"b{i}.attn.drop"]["mask"])
        blk.attn.proj_drop = Replay(saved[f"b{i}.proj_drop"]["mask"])
        blk.mlp.drop1 = Replay(saved[f"b{i}.mlp_drop"]["mask"])
        blk.mlp.drop2 = Replay(saved[f"b{i}.mlp2_drop"]["mask"])
        keys += [f"b{i}.attn.drop", f"b{i}.proj_drop", f"b{i}.mlp_drop", f"b{i}.mlp2_drop"]
    for k in keys:
        keep = saved[k]["mask"].float().mean().item()
        assert 0.6 < keep < 0.9, (k, keep)                        # keep rate ~ 0.75
    ref = oracle(x)
    torch.nn.functional.cross_entropy(ref, y).backward()
    assert _relerr(out.detach().cpu(), ref.detach()) < 1e-3
    po = dict(oracle.named_parameters())
    num = den = 0.0
    for name, prm in model.named_parameters():
        assert prm.grad is not None, name
        num += (prm.grad.detach().cpu().double() - po[name].grad.double()).pow(2).sum().item()
        den += po[name].grad.double().pow(2).sum().item()
    assert (num / den) ** 0.5 < 2e-3
    model.eval()
    with torch.no_grad():
        e1, e2 = model(x.to(DEV)), model(x.to(DEV))
    assert torch.equal(e1, e2)                                    # eval: every dropout is the identity


def test_unicom_drop_path_matches_oracle():
    """Stochastic depth of the unicom blocks (train mode, rate 0.1 per block): the HIP engine's per-sample keep draws
    are replayed in the oracle's DropPath modules; logits and gradients must then agree."""
    cfg_model = dict(model="unicom ViT-tiny-test", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    classes = ["a", "b", "c"]
    oracle, model = _pair(cfg_model, classes, seed=3)
    for net in (oracle, model):
        for blk in net.emb_model.blocks:
            blk.drop_path.drop_prob = 0.5          # make drops likely with 8 samples x 4 sites
    g = torch.Generator().manual_seed(11)
    x = torch.randn(8, 3, 56, 56, generator=g)
    y = torch.randint(0, 3, (8,), generator=g)
    oracle.train(); model.train()
    torch.manual_seed(21)
    out = model(x.to(DEV))
    eng = model._active
    dropped = 0
    for i, blk in enumerate(oracle.emb_model.blocks):
        s1, s2 = eng.saved[f"b{i}.dp1"]["scale"].cpu(), eng.saved[f"b{i}.dp2"]["scale"].cpu()
        assert set(s1.tolist()) <= {0.0, 2.0} and set(s2.tolist()) <= {0.0, 2.0}
        dropped += int((s1 == 0).sum() + (s2 == 0).sum())
        keeps = iter([(s1 > 0).float(), (s2 > 0).float()])
        # the two drop_path calls of a block share one module: feed the recorded draws in call order
        blk.drop_path.forward = (lambda mod, it: (lambda t: t * next(it).reshape(-1, 1, 1) / (1.0 - mod.drop_prob)))(blk.drop_path, keeps)
    assert 0 < dropped < 32
    ref = oracle(x)
    torch.nn.functional.cross_entropy(ref, y).backward()
    get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)(out, y.to(DEV)).backward()
    torch.cuda.synchronize()
    assert _relerr(out.detach().cpu(), ref.detach()) < 1e-3
    ref_p = dict(oracle.named_parameters())
    num = den = 0.0
    for name, p in model.named_parameters():
        num += (p.grad.cpu().double() - ref_p[name].grad.double()).pow(2).sum().item()
        den += ref_p[name].grad.double().pow(2).sum().item()
    assert (num / den) ** 0.5 < 3e-3, (num / den) ** 0.5
    # eval mode: no stochastic depth, batch statistics replaced by the running ones
    model.eval(); oracle.eval()
    for blk in oracle.emb_model.blocks:
        del blk.drop_path.forward
    with torch.no_grad():
        assert _relerr(model(x.to(DEV)).cpu(), oracle(x)) < 1e-3


@pytest.mark.parametrize("backbone", ["resnet_tiny_basic", "resnet_tiny_bottleneck"])
def test_eval_mode_folded_batchnorm_tracks_training(backbone):
    """Eval mode (val_epoch, engine.py:88-117) runs conv + folded BatchNorm + residual + ReLU as one launch per stage; the
    folded filters must follow the weights and running statistics through train steps and state-dict loads."""
    cfg_model = dict(model=backbone, pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    classes = ["a", "b", "c"]
    oracle, model = _pair(cfg_model, classes, seed=2)
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for n_, b_ in oracle.named_buffers():                      # non-trivial running statistics
            if n_.endswith("running_mean"):
                b_.copy_(torch.randn(b_.shape, generator=g) * 0.2)
            elif n_.endswith("running_var"):
                b_.copy_(torch.rand(b_.shape, generator=g) + 0.5)
        for p in oracle.parameters():
            if p.dim() == 1:
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.5)
    model.load_state_dict(oracle.state_dict())
    x = torch.randn(4, 3, 64, 64, generator=g)
    y = torch.randint(0, 3, (4,), generator=g)

    def check():
        oracle.eval(); model.eval()
        with torch.no_grad():
            ref, out = oracle(x), model(x.to(DEV)).cpu()
        assert _relerr(out, ref) < 1e-3 and out.argmax(-1).tolist() == ref.argmax(-1).tolist()
        return out

    first = check()
    # one SGD step on both sides: weights AND running statistics move; the next eval phase must see them
    o_opt = torch.optim.SGD(oracle.parameters(), lr=0.05)
    m_opt = get_optimizer(model, dict(type="sgd", lr=0.05))
    oracle.train(); model.train()
    torch.nn.functional.cross_entropy(oracle(x), y).backward(); o_opt.step()
    get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)(model(x.to(DEV)), y.to(DEV)).backward(); m_opt.step()
    second = check()
    assert (second - first).abs().max() > 1e-4
    # a state-dict load while in eval mode (model.py:170-172 checkpoint path)
    model.load_state_dict(OracleClassifier(cfg_model, classes).state_dict())
    oracle.load_state_dict(model.state_dict())
    third = check()
    assert (third - second).abs().max() > 1e-4


def test_classifier_dropout_train_path():
    """classifier_dropout > 0 (reference sample configs use 0.1): per-head masks, 1/(1-p) scaling, consistent backward."""
    from nkb_classification import hip
    cfg_model = dict(model="resnet_tiny_basic", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.4,
                     classifier_initialization="kaiming_normal_", task="multi")
    classes = {"a": ["x", "y"], "b": ["p", "q", "r"]}
    torch.manual_seed(1)
    model = get_model(cfg_model, classes, DEV)
    crit = get_loss(dict(task="multi", type="CrossEntropyLoss"), DEV)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(16, 3, 64, 64, generator=g).to(DEV)
    y = {"a": torch.randint(0, 2, (16,), generator=g), "b": torch.randint(0, 3, (16,), generator=g)}
    model.train()
    out = model(x)
    loss = crit(out, y)
    loss["loss"].backward()
    torch.cuda.synchronize()
    sv = model._active.saved["head"]
    emb = sv["emb"].float()
    for t, (name, head) in enumerate(model.classifier.items()):
        mask = sv["masks"][t].float()
        assert 0.45 < mask.mean().item() < 0.75                          # keep rate ~ 1 - p = 0.6
        dropped = sv["dropped"][t].float()
        torch.testing.assert_close(dropped, emb * mask / 0.6, rtol=1e-5, atol=1e-6)
        ref_logits = dropped @ head[1].weight.detach().t() + head[1].bias.detach()
        torch.testing.assert_close(out[name].detach(), ref_logits, rtol=1e-4, atol=1e-5)
        assert head[1].weight.grad is not None and torch.isfinite(head[1].weight.grad).all()
    assert not torch.equal(sv["masks"][0], sv["masks"][1])                # every head draws its own mask
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.emb_model.parameters())
    model.eval()
    with torch.no_grad():
        e1, e2 = model(x), model(x)
    assert torch.equal(e1["a"], e2["a"])                                  # eval: dropout is the identity


def test_grad_scaler_enabled_matches_unscaled_step():
    """cfg.enable_gradient_scaler=True (the reference's default): torch's GradScaler scales the loss, un-scales the arena
    gradient views in place and steps the fused optimizer; the update equals the un-scaled one."""
    cfg_model = dict(model="resnet_tiny_basic", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(8, 3, 64, 64, generator=g).to(DEV), torch.randint(0, 3, (8,), generator=g).to(DEV)
    results = []
    for enabled in (False, True):
        torch.manual_seed(0)
        model = get_model(cfg_model, ["a", "b", "c"], DEV)
        opt = get_optimizer(model, dict(type="adam", lr=1e-3))
        crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
        scaler = torch.amp.GradScaler("cuda", enabled=enabled, init_scale=1024.0)
        model.train()
        for _ in range(3):
            opt.zero_grad()
            loss = crit(model(x), y)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
        torch.cuda.synchronize()
        results.append(model.arena.flat_param.clone())
    assert torch.isfinite(results[1]).all()
    torch.testing.assert_close(results[1], results[0], rtol=2e-4, atol=2e-5)   # Adam normalises tiny gradients: 1-ulp scale/unscale differences move them


def test_multitask_bench_shape_runs_bf16():
    """BASELINE config 4 shape on one GPU: ResNet-50, 4 heads (2, 3, 5, 14 classes), focal loss gamma=1, bf16."""
    cfg_model = dict(model="resnet50", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="multi")
    classes = {"a": list("ab"), "b": list("abc"), "c": list("abcde"), "d": list("abcdefghijklmn")}
    torch.manual_seed(0)
    model = get_model(cfg_model, classes, DEV)
    opt = get_optimizer(model, dict(type="nadam", lr=1e-4, weight_decay=0.01))
    crit = get_loss(dict(task="multi", type="FocalLoss", gamma=1), DEV)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(32, 3, 224, 224, generator=g).to(DEV)
    y = {t: torch.randint(0, len(c), (32,), generator=g) for t, c in classes.items()}
    model.train()
    losses = []
    for _ in range(3):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x)
            loss = crit(out, y)
        loss["loss"].backward()
        opt.step()
        losses.append(loss["loss"].item())
    assert [tuple(out[t].shape) for t in classes] == [(32, 2), (32, 3), (32, 5), (32, 14)]
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0] * 1.5


def test_train_epoch_through_device_loader(tmp_path):
    """engine.train_epoch fed by get_dataset(device_pipeline=True): PNG folder -> uint8 canvases -> async H2D -> nkb_image_prep
    -> the HIP model; the batches the engine sees equal the host-side float pipeline of the same folder."""
    import numpy as np
    from PIL import Image
    from nkb_classification.dataset import get_dataset
    rng = np.random.default_rng(1)
    for cls, n in (("a", 5), ("b", 3)):
        (tmp_path / cls).mkdir()
        for i in range(n):
            Image.fromarray(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)).save(tmp_path / cls / f"{i}.png")
    base = dict(root=str(tmp_path), size=64, batch_size=4, num_workers=0, shuffle=False)
    dev_loader = get_dataset(dict(base, device_pipeline=True, device=DEV))
    host_loader = get_dataset(dict(base))
    assert dev_loader.dataset.classes == ["a", "b"] and len(dev_loader) == len(host_loader) == 2
    for (xd, yd), (xh, yh) in zip(dev_loader, host_loader):
        assert xd.is_cuda and yd.is_cuda and yd.cpu().tolist() == yh.tolist()
        torch.testing.assert_close(xd.cpu(), xh, rtol=1e-5, atol=1e-5)        # square images: no resize / pad difference
    cfg_model = dict(model="resnet_tiny_basic", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task="single")
    torch.manual_seed(0)
    model = get_model(cfg_model, ["a", "b"], DEV)
    opt = get_optimizer(model, dict(type="sgd", lr=0.01))
    crit = get_loss(dict(task="single", type="CrossEntropyLoss"), DEV)
    logger = BaseLogger(_cfg("single"), ["a", "b"])
    res = train_epoch(model, dev_loader, opt, None, torch.amp.GradScaler("cuda", enabled=False), crit, DEV, _cfg("single"), logger)
    assert len(res["running_loss"]) == 2 and all(math.isfinite(v) for v in res["running_loss"])
    assert len(res["predictions"]) == 8
