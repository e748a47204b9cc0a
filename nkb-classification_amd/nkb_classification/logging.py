"""Per-iteration result accumulation — drop-in for the hot-path part of
/root/reference/nkb_classification/logging.py (BaseLogger, lines 218-294).

The reference converts predictions to python lists with four blocking device->host copies per step
(logging.py:268-281).  Here the per-step tensors stay on the device (the softmax / argmax come for free from
the fused loss kernel, see losses._LossFn) and are converted once in `get_epoch_results()`, which returns the
same dict of python lists (`running_loss`, `confidences`, `predictions`, `ground_truth`, `images`).

The CSV / Comet sinks of the reference (LocalExperiment, TrainLogger.log_epoch, ...) are I/O glue outside the
hot path and are not reproduced here; `LocalExperiment` keeps the metrics.csv contract train.py relies on.
"""
from __future__ import annotations

from collections import defaultdict
from collections.abc import Sequence
from pathlib import Path

import numpy as np
import torch

from . import hip
from .utils import save_classes


def softmax_argmax(logits: torch.Tensor):
    """softmax(dim=-1, fp32) and argmax of a [B, C] fp32 logits tensor via the fused HIP kernel."""
    side = getattr(logits, "_nkb_side", None)
    if side is not None:
        return side
    hip.require_device(logits, "log_iter")
    x = logits.detach()
    if x.dtype != torch.float32 or x.stride(-1) != 1:
        x = x.float().contiguous()
    B, C = x.shape
    probs = torch.empty(B, C, device=x.device, dtype=torch.float32)
    am = torch.empty(B, device=x.device, dtype=torch.int32)
    hip.loss_forward(0, x, x.stride(0), None, B, C, None, 0.0, -100, probs, C, am, None, None)
    return probs, am


class BaseLogger:
    def __init__(self, cfg, classes):
        assert cfg.task in ("single", "multi")
        self.cfg = cfg
        self.task = cfg.task
        self.classes = classes
        # logging.py:243 reads an attribute that is never set (multi-task construction crashes at reference HEAD);
        # the evident intent — sorted task names — is what is implemented here.
        self.target_names = None if self.task == "single" else sorted(classes)
        self.init_iter_logs()

    def init_iter_logs(self):
        self.epoch_images_example = None
        mk = list if self.task == "single" else (lambda: defaultdict(list))
        self._loss, self._conf, self._pred, self._gt = mk(), mk(), mk(), mk()

    def log_iter(self, pred, true, loss):
        assert type(pred) == type(true)
        if isinstance(pred, dict):
            assert pred.keys() == true.keys()
            for name in pred.keys():
                probs, am = softmax_argmax(pred[name])
                self._gt[name].append(true[name])
                self._conf[name].append(probs)
                self._pred[name].append(am)
                self._loss[name].append(loss[name].detach())
            self._loss["loss"].append(loss["loss"].detach())
        else:
            probs, am = softmax_argmax(pred)
            self._gt.append(true)
            self._conf.append(probs)
            self._pred.append(am)
            self._loss.append(loss.detach())

    def log_images_if_needed(self, images):
        if self.epoch_images_example is None:
            self.epoch_images_example = images.to("cpu")

    @staticmethod
    def _rows(chunks):
        return torch.cat([c.detach().reshape(c.shape[0], -1) if c.dim() > 1 else c.detach() for c in chunks]).cpu() \
            .numpy().tolist() if chunks else []

    @staticmethod
    def _scalars(vals):
        return torch.stack([v.reshape(()) for v in vals]).float().cpu().numpy().tolist() if vals else []

    def get_epoch_results(self):
        if self.task == "single":
            loss, conf = self._scalars(self._loss), self._rows(self._conf)
            pred, gt = self._rows(self._pred), self._rows([g.to(torch.int64) for g in self._gt])
        else:
            loss, conf, pred, gt = defaultdict(list), defaultdict(list), defaultdict(list), defaultdict(list)
            for k, v in self._loss.items():
                loss[k] = self._scalars(v)
            for k in self._conf:
                conf[k] = self._rows(self._conf[k])
                pred[k] = self._rows(self._pred[k])
                gt[k] = self._rows([g.to(torch.int64) for g in self._gt[k]])
        self.epoch_running_loss, self.epoch_confidences = loss, conf
        self.epoch_predictions, self.epoch_ground_truth = pred, gt
        return {
            "running_loss": loss,
            "confidences": conf,
            "predictions": pred,
            "ground_truth": gt,
            "images": self.epoch_images_example,
        }


class LocalExperiment:
    """Scalar-metric sink of a run directory.  File contract of the reference's writer (logging.py:18-38), which the eval /
    plotting tooling around it reads: `<path>/metrics.csv`, tab-separated, header `Epoch` followed by the metric names in
    sorted order, one row per epoch, missing values left empty, file rewritten whenever a value arrives."""

    def __init__(self, path=""):
        self.path = Path(path)
        self._rows = {}          # epoch -> {column: value}

    def log_metric(self, name, value, epoch=0, step=None, prefix=None):
        column = name if prefix is None else f"{prefix}/{name}"
        self._rows.setdefault(epoch, {})[column] = float(np.mean(value)) if isinstance(value, Sequence) else value
        self._write()

    def log_metrics(self, metrics_dict, epoch=0, step=None, prefix=None):
        for name, value in metrics_dict.items():
            self._rows.setdefault(epoch, {})[name if prefix is None else f"{prefix}/{name}"] = \
                float(np.mean(value)) if isinstance(value, Sequence) else value
        self._write()

    def _write(self):
        columns = sorted({c for row in self._rows.values() for c in row})
        lines = ["\t".join(["Epoch"] + columns)]
        for i, epoch in enumerate(sorted(self._rows)):
            row = self._rows[epoch]
            lines.append("\t".join([str(i)] + ["" if row.get(c) is None else repr(float(row[c])) for c in columns]))
        (self.path / "metrics.csv").write_text("\n".join(lines) + "\n")


def get_local_experiment(cfg_exp):
    """Fresh run directory `<path>`, `<path>1`, `<path>2`, ... with a `weights/` sub-directory (logging.py:56-66).  Called by
    rank 0 only under data parallelism (train.py)."""
    assert cfg_exp is not None and "path" in cfg_exp.keys()
    base, n = cfg_exp["path"], 0
    while True:
        run = Path(base if n == 0 else f"{base}{n}")
        try:
            run.mkdir(parents=True, exist_ok=False)
            (run / "weights").mkdir()
            return LocalExperiment(run)
        except FileExistsError:
            n += 1


class TrainLogger(BaseLogger):
    """Epoch-level sink used by train.train(): classes.json + scalar metrics into metrics.csv."""

    def __init__(self, cfg, comet_experiment, local_experiment, classes):
        super().__init__(cfg, classes)
        self.comet_experiment = comet_experiment
        self.local_experiment = local_experiment
        if local_experiment is not None:
            save_classes(classes, local_experiment.path / "classes.json")

    def log_images_at_start(self, loader):
        return None

    def log_epoch(self, epoch, train_results, val_results):
        if self.local_experiment is None:
            return
        for fold, res in (("Train", train_results), ("Validation", val_results)):
            m = res["metrics"]
            if self.task == "single":
                scal = {"loss": m["epoch_loss"], "balanced accuracy": m["epoch_acc"]}
            else:
                scal = {"balanced accuracy": m["epoch_acc"], "loss": float(np.mean(m["loss"]))}
                for t in self.cfg.target_names:
                    scal[f"{t} balanced accuracy"] = m[t]["epoch_acc"]
                    scal[f"{t} loss"] = m[t]["epoch_loss"]
            self.local_experiment.log_metrics(scal, epoch=epoch, prefix=fold)
