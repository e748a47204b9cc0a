"""train.py on a multi-task synthetic config (4 heads, FocalLoss gamma 1, gradient logging on, scaler on, backbone dropout), twice:
identical final state and metrics.  The single-task twin is scripts/soak_train.py."""
import os, subprocess, sys, tempfile
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
src = (ROOT / "nkb-classification_amd" / "configs" / "synthetic_singletask_config.py").read_text()
edits = [("n_epochs = 2", f"n_epochs = {epochs}"), ('task = "single"', 'task = "multi"'),
         ("classes = [str(i) for i in range(10)]", 'classes = {"a": list("ab"), "b": list("abc"), "c": list("abcde"), "d": list("abcdefghijklmn")}\ntarget_names = sorted(classes)    # configs/multitask_config.py:51'),
         ('"model": "resnet18"', '"model": "resnet50"'), ('"classifier_dropout": 0.0', '"classifier_dropout": 0.1'), ('"backbone_dropout": 0.0', '"backbone_dropout": 0.1'),
         ("enable_gradient_scaler = False", "enable_gradient_scaler = True"), ("log_gradients = False", "log_gradients = True"),
         ("show_full_current_loss_in_terminal = False", "show_full_current_loss_in_terminal = True"),
         ('"type": "CrossEntropyLoss"', '"type": "FocalLoss", "gamma": 1'), ('"n_images": 256', '"n_images": 384')]
for a, b in edits:
    assert a in src, a
    src = src.replace(a, b)
outs = []
for k in range(2):
    d = Path(tempfile.mkdtemp(prefix=f"soak_train_multi_{k}_"))
    (d / "cfg.py").write_text(src.replace('"runs/synthetic_single"', repr(str(d / "exp"))))
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "ddp_train_probe.py"), "-cfg", str(d / "cfg.py")], capture_output=True, text=True,
                       env=dict(os.environ, NKB_DUMP_PARAMS=str(d)))
    if r.returncode != 0:
        print(r.stderr[-4000:]); sys.exit(2)
    outs.append((torch.load(d / "params_rank0.pt"), (d / "exp" / "metrics.csv").read_text()))
(a, ma), (b, mb) = outs
same = torch.equal(a["flat_param"], b["flat_param"]) and all(torch.equal(a["buffers"][k], b["buffers"][k]) for k in a["buffers"]) and ma == mb
print(ma)
print(f"multi-task resnet50, {epochs} epochs twice: finite {bool(torch.isfinite(a['flat_param']).all())}, identical parameters / buffers / metrics: {same}")
sys.exit(0 if same and torch.isfinite(a["flat_param"]).all() else 1)
