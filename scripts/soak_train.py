"""Soak of the whole config-driven path: nkb-classification_amd/train.py (through tests/ddp_train_probe.py, which dumps the final
state) run TWICE in separate processes on a synthetic config that exercises the epoch driver — ResNet-50, several epochs, gradient
scaler on, classifier dropout, a freeze -> unfreeze backbone policy, validation every epoch, checkpoints — and compared: parameters,
buffers and metrics.csv must be identical (same seed => same run).  Usage: python scripts/soak_train.py [epochs] [model] [ranks]"""
import os, subprocess, sys, tempfile
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
model = sys.argv[2] if len(sys.argv) > 2 else "resnet50"
ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # > 1: torch.distributed.run, gloo transport, every rank on this GPU
src = (ROOT / "nkb-classification_amd" / "configs" / "synthetic_singletask_config.py").read_text()
edits = [("n_epochs = 2", f"n_epochs = {epochs}"), ('"model": "resnet18"', f'"model": "{model}"'), ('"classifier_dropout": 0.0', '"classifier_dropout": 0.2'),
         ("enable_gradient_scaler = False", "enable_gradient_scaler = True"), ('backbone_state_policy = {0: "unfreeze"}', 'backbone_state_policy = {0: "freeze", 2: "unfreeze"}'),
         ('"n_images": 256', '"n_images": 640')]
for a, b in edits:
    assert a in src, a
    src = src.replace(a, b)
outs = []
for k in range(2):
    d = Path(tempfile.mkdtemp(prefix=f"soak_train_{k}_"))
    (d / "cfg.py").write_text(src.replace('"runs/synthetic_single"', repr(str(d / "exp"))))
    env = dict(os.environ, NKB_DUMP_PARAMS=str(d))
    cmd = [sys.executable, str(ROOT / "tests" / "ddp_train_probe.py"), "-cfg", str(d / "cfg.py")]
    if ranks > 1:
        env.update(NKB_DDP_BACKEND="gloo", NKB_DDP_ONE_GPU="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(29700 + os.getpid() % 200 + k)] + cmd[1:]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    if r.returncode != 0:
        print(r.stderr[-3000:]); sys.exit(2)
    outs.append((torch.load(d / "params_rank0.pt"), (d / "exp" / "metrics.csv").read_text()))
(a, ma), (b, mb) = outs
same = torch.equal(a["flat_param"], b["flat_param"]) and all(torch.equal(a["buffers"][k], b["buffers"][k]) for k in a["buffers"]) and ma == mb
print(ma)
print(f"{model}, {epochs} epochs twice on {ranks} rank(s): finite {bool(torch.isfinite(a['flat_param']).all())}, identical parameters / buffers / metrics: {same}")
sys.exit(0 if same and torch.isfinite(a["flat_param"]).all() else 1)
