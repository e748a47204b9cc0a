"""Epoch metrics — same sklearn calls as /root/reference/nkb_classification/metrics.py:7-70
(balanced accuracy, per-class ROC-AUC, mean loss).  Epoch-end host code, O(N); not a kernel target."""
from __future__ import annotations

import warnings

import numpy as np
from sklearn.metrics import balanced_accuracy_score, roc_auc_score
from sklearn.preprocessing import label_binarize


def compute_targetwise_metrics(epoch_results, target_name=None):
    pick = (lambda k: epoch_results[k]) if target_name is None else (lambda k: epoch_results[k][target_name])
    running_loss, predictions, ground_truth = pick("running_loss"), pick("predictions"), pick("ground_truth")
    confidences = np.array(pick("confidences"))
    n_classes = confidences.shape[1]
    present = np.unique(ground_truth)
    if len(present) < n_classes:
        warnings.warn("\nNumber of classes in ground truth is less than number of classes in predicted "
                      "confidences. \nSome of ROC AUC metric values will be NaN\n")
    acc = balanced_accuracy_score(ground_truth, predictions)
    if n_classes > 2:
        auc = np.full(n_classes, np.nan)
        if len(present) > 1:
            onehot = label_binarize(ground_truth, classes=range(n_classes))
            for c in present:
                auc[c] = roc_auc_score(onehot[:, c], confidences[:, c])
    else:
        auc = np.nan
        if len(present) > 1:
            auc = roc_auc_score(ground_truth, confidences[:, 1])
    return {"epoch_acc": acc, "epoch_roc_auc": auc, "epoch_loss": np.mean(running_loss)}


def compute_metrics(cfg, epoch_results: dict):
    if cfg.task == "single":
        metrics = compute_targetwise_metrics(epoch_results)
        metrics["loss"] = epoch_results["running_loss"]
        return metrics
    if cfg.task == "multi":
        names = cfg.target_names
        metrics = {t: compute_targetwise_metrics(epoch_results, t) for t in names}
        metrics["loss"] = epoch_results["running_loss"]["loss"]
        metrics["epoch_acc"] = np.mean([metrics[t]["epoch_acc"] for t in names])
        return metrics
    raise ValueError(f"Unknown task type {cfg.task} for metric computation")
