"""ORACLE fixture generator (test infrastructure; runs only where /root/reference is mounted).

Imports the reference's own importable modules
(nkb_classification.{engine,losses,metrics,utils}; model/logging/dataset need
timm/comet_ml/albumentations, which are not installed) and records golden
input/output vectors G1-G5 of SURVEY.md §8(c) as JSON under tests/golden/.
Only the generated DATA files travel to the GPU box; this script is inert when
the reference tree is absent.

    python oracle/make_golden.py
"""
from __future__ import annotations

import json
import math
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"


def _tolist(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().tolist()
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    if isinstance(x, dict):
        return {k: _tolist(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_tolist(v) for v in x]
    return x


def g1_losses(ref_losses):
    x = torch.tensor([[2, -1, .5], [.1, .2, .3], [-3, 4, 0], [1, 1, 1]], dtype=torch.float32)
    y = torch.tensor([0, 2, 1, 1])
    cases = []

    def run(cfg, xx, yy, name):
        crit = ref_losses.get_loss(cfg, "cpu")
        if isinstance(xx, dict):
            xs = {k: v.clone().requires_grad_(True) for k, v in xx.items()}
            out = crit(xs, yy)
            out["loss"].backward()
            cases.append(dict(name=name, cfg=cfg, x=_tolist(xx), y=_tolist(yy),
                              loss={k: v.item() for k, v in out.items()},
                              grad={k: _tolist(v.grad) for k, v in xs.items()}))
        else:
            xs = xx.clone().requires_grad_(True)
            out = crit(xs, yy)
            if out.requires_grad:
                out.backward()
            cases.append(dict(name=name, cfg=cfg, x=_tolist(xx), y=_tolist(yy), loss=out.item(),
                              grad=_tolist(xs.grad) if xs.grad is not None else None))

    run(dict(task="single", type="CrossEntropyLoss"), x, y, "ce")
    run(dict(task="single", type="CrossEntropyLoss", weight=[1, 2, .5]), x, y, "ce_weighted")
    run(dict(task="single", type="FocalLoss"), x, y, "focal_g2")
    run(dict(task="single", type="FocalLoss", gamma=1, alpha=[1, 2, .5]), x, y, "focal_g1_alpha")
    run(dict(task="single", type="FocalLoss"), x, torch.full((4,), -100), "focal_all_ignored")
    run(dict(task="multi", type="FocalLoss", gamma=1),
        {"a": x[:, :2].contiguous(), "b": x}, {"a": torch.tensor([0, 1, 1, 0]), "b": y}, "multi_focal_g1")
    run(dict(task="multi", type="CrossEntropyLoss"),
        {"a": x[:, :2].contiguous(), "b": x}, {"a": torch.tensor([0, 1, 1, 0]), "b": y}, "multi_ce")
    g = torch.Generator().manual_seed(7)
    for C in (2, 5, 14, 1000):
        rows = 32 if C < 100 else 4
        xr = torch.randn(rows, C, generator=g) * 3
        yr = torch.randint(0, C, (rows,), generator=g)
        run(dict(task="single", type="CrossEntropyLoss"), xr, yr, f"ce_rand_C{C}")
        run(dict(task="single", type="FocalLoss", gamma=1), xr, yr, f"focal_g1_rand_C{C}")
    w5 = [0.5, 1.0, 2.0, 1.5, 0.25]
    xr = torch.randn(32, 5, generator=g) * 3
    yr = torch.randint(0, 5, (32,), generator=g)
    run(dict(task="single", type="CrossEntropyLoss", weight=w5), xr, yr, "ce_weighted_rand_C5")
    run(dict(task="single", type="FocalLoss", gamma=2, alpha=w5), xr, yr, "focal_g2_alpha_rand_C5")
    try:
        ref_losses.get_loss(dict(task="single", type="Nope"), "cpu")
        err = None
    except NotImplementedError as e:
        err = str(e)
    return dict(cases=cases, unknown_type_error=err)


class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.emb_model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4))
        self.classifier = torch.nn.Sequential(torch.nn.Dropout(0.0), torch.nn.Linear(4, 3))

    def forward(self, x):
        return self.classifier(self.emb_model(x))


def g2_optim(ref_utils):
    out = dict(trajectories=[], lr_sequences={}, group_defaults={})
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(8, 6, generator=g) for _ in range(6)]
    ys = [torch.randint(0, 3, (8,), generator=g) for _ in range(6)]
    cfgs = {
        "nadam": dict(type="nadam", lr=1e-2, backbone_lr=5e-3, classifier_lr=2e-2, weight_decay=0.01,
                      backbone_weight_decay=0.01, classifier_weight_decay=0.2),
        "adam": dict(type="adam", lr=1e-2, weight_decay=0.05, classifier_lr=3e-2),
        "radam": dict(type="radam", lr=1e-2, backbone_weight_decay=0.1),
        "sgd": dict(type="sgd", lr=5e-2, classifier_weight_decay=0.3),
    }
    for name, cfg in cfgs.items():
        m = _Tiny()
        init = {k: _tolist(v) for k, v in m.state_dict().items()}
        opt = ref_utils.get_optimizer(m, cfg)
        out["group_defaults"][name] = [
            {k: (list(v) if isinstance(v, tuple) else v) for k, v in grp.items()
             if k != "params" and isinstance(v, (int, float, bool, tuple, type(None)))}
            for grp in opt.param_groups
        ]
        steps = []
        for x, y in zip(xs, ys):
            opt.zero_grad()
            torch.nn.functional.cross_entropy(m(x), y).backward()
            grads = {k: _tolist(p.grad) for k, p in m.named_parameters()}
            opt.step()
            steps.append(dict(grads=grads, params={k: _tolist(v) for k, v in m.state_dict().items()}))
        out["trajectories"].append(dict(name=name, cfg=cfg, init=init, x=_tolist(xs), y=_tolist(ys), steps=steps))
    for name, pol in {
        "step": dict(type="step", step_size=2, gamma=0.5),
        "multistep": dict(type="multistep", steps=[1, 3], gamma=0.1),
        "cosine": dict(type="cosine", n_epochs=4),
    }.items():
        m = _Tiny()
        opt = ref_utils.get_optimizer(m, dict(type="sgd", lr=1.0))
        sch = ref_utils.get_scheduler(opt, pol)
        seq = []
        for _ in range(5):
            seq.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out["lr_sequences"][name] = dict(policy=pol, lrs=seq)
    m = _Tiny()
    out["empty_policy_is_none"] = ref_utils.get_scheduler(ref_utils.get_optimizer(m, dict(type="sgd")), {}) is None
    for bad, fn in (("optimizer", lambda: ref_utils.get_optimizer(m, dict(type="lion"))),
                    ("scheduler", lambda: ref_utils.get_scheduler(ref_utils.get_optimizer(m, dict(type="sgd")),
                                                                  dict(type="poly")))):
        try:
            fn()
            out[f"unknown_{bad}_error"] = None
        except NotImplementedError as e:
            out[f"unknown_{bad}_error"] = str(e)
    return out


def g3_metrics(ref_metrics):
    rng = np.random.default_rng(5)
    out = {}

    def one(n, C):
        logits = rng.normal(size=(n, C))
        conf = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
        return dict(running_loss=rng.random(4).tolist(), confidences=conf.tolist(),
                    predictions=conf.argmax(1).tolist(), ground_truth=rng.integers(0, C, n).tolist())

    single2, single5 = one(24, 2), one(40, 5)
    for name, res in (("single_C2", single2), ("single_C5", single5)):
        cfg = types.SimpleNamespace(task="single")
        met = ref_metrics.compute_metrics(cfg, dict(res))
        out[name] = dict(inputs=res, metrics=_tolist(met))
    a, b = one(30, 2), one(30, 3)
    multi = {k: {"a": a[k], "b": b[k]} for k in ("running_loss", "confidences", "predictions", "ground_truth")}
    multi["running_loss"]["loss"] = (np.array(a["running_loss"]) + np.array(b["running_loss"])).tolist()
    cfg = types.SimpleNamespace(task="multi", target_names=["a", "b"])
    out["multi"] = dict(inputs=multi, metrics=_tolist(ref_metrics.compute_metrics(cfg, multi)))
    return out


class _LoggerAdapter:
    """Duck-typed epoch_logger (logging.py:245-294 method names) over the oracle's EpochLog."""

    def __init__(self, multi):
        from oracle.torch_engine import EpochLog
        self._log = EpochLog(multi)

    def init_iter_logs(self):
        self._log.reset()

    def log_iter(self, pred, true, loss):
        self._log.add(pred, true, loss)

    def log_images_if_needed(self, images):
        if self._log.images is None:
            self._log.images = images

    def get_epoch_results(self):
        return self._log.results()


def _engine_case(ref_engine, ref_losses, ref_utils, *, backbone, classes, hw, n_images, batch, crit_cfg,
                 opt_cfg, n_epochs_cos, log_gradients, task, epochs=1, seed=0):
    from oracle.torch_models import OracleClassifier
    from oracle.torch_engine import synthetic_batches
    torch.manual_seed(seed)
    cfg_model = dict(model=backbone, pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                     classifier_initialization="kaiming_normal_", task=task)
    model = OracleClassifier(cfg_model, classes)
    n_cls = {t: len(c) for t, c in classes.items()} if isinstance(classes, dict) else len(classes)
    train = synthetic_batches(n_images, batch, n_cls, seed=1234, hw=hw)
    val = synthetic_batches(2 * batch, batch, n_cls, seed=4321, hw=hw)
    opt = ref_utils.get_optimizer(model, opt_cfg)
    sch = ref_utils.get_scheduler(opt, dict(type="cosine", n_epochs=n_epochs_cos))
    crit = ref_losses.get_loss(crit_cfg, "cpu")
    cfg = types.SimpleNamespace(task=task, enable_mixed_presicion=False, log_gradients=log_gradients,
                                show_full_current_loss_in_terminal=False)
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    logger = _LoggerAdapter(task == "multi")
    rec = dict(cfg_model=cfg_model, classes=classes, hw=hw, n_images=n_images, batch=batch, crit_cfg=crit_cfg,
               opt_cfg=opt_cfg, n_epochs_cos=n_epochs_cos, seed=seed, epochs=[])
    for _ in range(epochs):
        tr = ref_engine.train_epoch(model, train, opt, sch, scaler, crit, "cpu", cfg, logger)
        ep = dict(train_running_loss=_tolist(tr["running_loss"]), train_predictions=_tolist(tr["predictions"]),
                  train_ground_truth=_tolist(tr["ground_truth"]),
                  lr_after=[g["lr"] for g in opt.param_groups])
        if log_gradients:
            ep["grad_total"] = [float(v) for v in tr["metrics_grad_log"]["Gradients/Total"]]
            ep["grad_keys"] = sorted(tr["metrics_grad_log"].keys())[:6]
            ep["grad_first_step"] = {k: float(v[0]) for k, v in tr["metrics_grad_log"].items()}
        va = ref_engine.val_epoch(model, val, crit, "cpu", cfg, logger)
        ep.update(val_running_loss=_tolist(va["running_loss"]), val_predictions=_tolist(va["predictions"]),
                  val_confidences=_tolist(va["confidences"]))
        rec["epochs"].append(ep)
    model.eval()
    with torch.no_grad():
        out = model(val[0][0])
    rec["final_val_logits"] = _tolist(out)
    rec["final_val_argmax"] = _tolist({t: v.argmax(-1) for t, v in out.items()} if isinstance(out, dict)
                                      else out.argmax(-1))
    rec["param_norms"] = {k: float(v.float().norm()) for k, v in model.state_dict().items()}
    return rec


def g4_engine(ref_engine, ref_losses, ref_utils):
    nadam = dict(type="nadam", lr=1e-3, backbone_lr=1e-4, classifier_lr=1e-3, weight_decay=0.01,
                 backbone_weight_decay=0.01, classifier_weight_decay=0.2)
    out = {}
    # BASELINE config 1: ResNet-18, 2 classes, 64 synthetic 224x224 images, bs=8, fp32, CPU
    out["config1_resnet18"] = _engine_case(
        ref_engine, ref_losses, ref_utils, backbone="resnet18", classes=["a", "b"], hw=224, n_images=64, batch=8,
        crit_cfg=dict(task="single", type="CrossEntropyLoss"), opt_cfg=nadam, n_epochs_cos=5,
        log_gradients=True, task="single")
    # small members of the same families for fast parity tests
    out["tiny_basic_single"] = _engine_case(
        ref_engine, ref_losses, ref_utils, backbone="resnet_tiny_basic", classes=["a", "b", "c"], hw=64, n_images=32,
        batch=8, crit_cfg=dict(task="single", type="FocalLoss", gamma=2), opt_cfg=dict(type="adam", lr=1e-3),
        n_epochs_cos=4, log_gradients=False, task="single", epochs=2)
    out["tiny_bottleneck_multi"] = _engine_case(
        ref_engine, ref_losses, ref_utils, backbone="resnet_tiny_bottleneck",
        classes={"color": ["r", "g"], "shape": ["a", "b", "c"], "size": list("12345"), "kind": list("abcdefghijklmn")},
        hw=64, n_images=32, batch=8, crit_cfg=dict(task="multi", type="FocalLoss", gamma=1), opt_cfg=nadam,
        n_epochs_cos=4, log_gradients=False, task="multi", epochs=2)
    out["tiny_vit_single"] = _engine_case(
        ref_engine, ref_losses, ref_utils, backbone="vit_tiny_test", classes=["a", "b", "c", "d"], hw=64, n_images=32,
        batch=8, crit_cfg=dict(task="single", type="CrossEntropyLoss"), opt_cfg=dict(type="sgd", lr=1e-2),
        n_epochs_cos=4, log_gradients=False, task="single", epochs=2)
    return out


def _import_ref_logging():
    """logging.py imports comet_ml / torchvision / matplotlib at module level only for its Comet and image-grid sinks; none of
    them is installed here.  Empty stand-in modules satisfy those imports; BaseLogger itself — the code
    the goldens come from — uses none of them."""
    import importlib
    import types
    for name, attrs in (("comet_ml", {"Experiment": object}), ("torchvision", {}), ("torchvision.transforms", {}),
                        ("torchvision.utils", {"make_grid": None}), ("matplotlib", {}), ("matplotlib.pyplot", {})):
        try:
            importlib.import_module(name)
        except Exception:
            mod = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(mod, k, v)
            sys.modules[name] = mod
            if "." in name:
                setattr(sys.modules[name.split(".")[0]], name.split(".")[1], mod)
    from nkb_classification import logging as ref_logging
    assert str(REF) in ref_logging.__file__
    return ref_logging


def g5_logger():
    """BaseLogger.init_iter_logs / log_iter / get_epoch_results (logging.py:245-294) run on fixed inputs: the probe rows of
    SURVEY.md §8(c), a seeded multi-batch single-task case, and a multi-task case.  The multi-task CONSTRUCTOR reads an
    attribute that is never assigned (logging.py:243) and cannot run at reference HEAD, so that instance is created without
    it and its four attributes set by hand; everything recorded below is then produced by the reference's own methods."""
    ref_logging = _import_ref_logging()
    from types import SimpleNamespace
    out = {}
    lg = ref_logging.BaseLogger(SimpleNamespace(task="single"), ["a", "b"])
    lg.init_iter_logs()
    preds = torch.tensor([[.1, .9], [2., -1.]])
    lg.log_iter(preds, torch.tensor([1, 1]), torch.tensor(0.75))
    res = lg.get_epoch_results()
    out["probe"] = dict(preds=preds.tolist(), true=[1, 1], loss=0.75, running_loss=res["running_loss"],
                        confidences=res["confidences"], predictions=res["predictions"], ground_truth=res["ground_truth"])
    g = torch.Generator().manual_seed(5)
    lg.init_iter_logs()
    batches = []
    for b in (8, 8, 5):
        x = torch.randn(b, 7, generator=g) * 3
        y = torch.randint(0, 7, (b,), generator=g)
        loss = torch.rand((), generator=g)
        lg.log_iter(x, y, loss)
        batches.append(dict(preds=x.tolist(), true=y.tolist(), loss=float(loss)))
    lg.log_images_if_needed(torch.zeros(1, 3, 2, 2))
    res = lg.get_epoch_results()
    out["single"] = dict(batches=batches, running_loss=res["running_loss"], confidences=res["confidences"],
                         predictions=res["predictions"], ground_truth=res["ground_truth"],
                         images_shape=list(res["images"].shape))
    classes = {"shape": ["a", "b", "c"], "color": ["r", "g"]}
    mt = ref_logging.BaseLogger.__new__(ref_logging.BaseLogger)
    mt.cfg, mt.task, mt.classes, mt.target_names = SimpleNamespace(task="multi"), "multi", classes, sorted(classes)
    mt.init_iter_logs()
    batches = []
    for b in (4, 3):
        pred = {t: torch.randn(b, len(c), generator=g) for t, c in classes.items()}
        true = {t: torch.randint(0, len(c), (b,), generator=g) for t, c in classes.items()}
        loss = {t: torch.rand((), generator=g) for t in classes}
        loss["loss"] = sum(loss.values())
        mt.log_iter(pred, true, loss)
        batches.append(dict(preds={t: v.tolist() for t, v in pred.items()}, true={t: v.tolist() for t, v in true.items()},
                            loss={t: float(v) for t, v in loss.items()}))
    res = mt.get_epoch_results()
    out["multi"] = dict(classes=classes, batches=batches, running_loss=dict(res["running_loss"]),
                        confidences=dict(res["confidences"]), predictions=dict(res["predictions"]),
                        ground_truth=dict(res["ground_truth"]))
    out["source"] = "reference BaseLogger (logging.py:218-294) imported with stand-in modules for comet_ml / torchvision / matplotlib"
    return out


def main():
    if not REF.exists():
        print("reference tree absent: nothing to do")
        return 0
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(ROOT))
    from nkb_classification import engine as ref_engine, losses as ref_losses, metrics as ref_metrics, \
        utils as ref_utils
    assert str(REF) in ref_engine.__file__
    torch.set_num_threads(8)
    OUT.mkdir(parents=True, exist_ok=True)
    makers = dict(g1_losses=lambda: g1_losses(ref_losses), g2_optim=lambda: g2_optim(ref_utils),
                  g3_metrics=lambda: g3_metrics(ref_metrics),
                  g4_engine=lambda: g4_engine(ref_engine, ref_losses, ref_utils), g5_logger=g5_logger)
    only = sys.argv[1:]                     # e.g. `python oracle/make_golden.py g5_logger` regenerates one fixture
    blobs = {name: make() for name, make in makers.items() if not only or name in only}
    for name, blob in blobs.items():
        blob["_meta"] = dict(torch=torch.__version__, generator="oracle/make_golden.py",
                             reference="nkb-tech/nkb-classification @ /root/reference")
        (OUT / f"{name}.json").write_text(json.dumps(blob))
        print(name, (OUT / f"{name}.json").stat().st_size, "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
