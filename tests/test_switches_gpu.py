"""Every NKB_* switch that still selects an algorithmic path at run time, at its NON-default value, on a train step next to the
oracle: the default combination is what every other test runs; this file is the proof that the other side of each
remaining switch still computes the same step.  One child process per value (tests/switch_probe.py: the switches are read once
per process).  ResNet-50 at batch 64 / 128 x 128 px reaches every ResNet-side path (Gram-form closing stages, the row-balanced
3x3 core from 4 096 pixels up, the eight-phase GEMM envelope is not reached at this size and keeps its own tests); the transformer
switches run on unicom ViT-B/32.  The probe's bars are those of tests/test_parity_bench_size_gpu.py.
The set is closed: test_switch_surface_is_what_this_file_covers counts the names the sources read."""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]

RESNET = ("resnet50", "64", "bf16", "128")
UNICOM = ("unicom ViT-B/32", "16", "bf16", "224")
UNICOM8 = ("unicom ViT-B/32", "16", "fp8", "224")
CASES = [
    ("NKB_GRAM_BN", "0", RESNET), ("NKB_GRAM_MAX_C", "256", RESNET), ("NKB_CONVP", "0", RESNET), ("NKB_WGRAD_STREAM", "0", RESNET),
    ("NKB_DET_WGRAD", "0", RESNET), ("NKB_FUSED_BNBWD", "0", RESNET), ("NKB_FUSED_RES_BNBWD", "0", RESNET), ("NKB_RELU_BITS", "0", RESNET),
    ("NKB_S2_CLASSES", "0", RESNET), ("NKB_HALO", "0", RESNET), ("NKB_PACKED_STEM", "0", RESNET), ("NKB_WGRAD3X3", "0", RESNET),
    ("NKB_WGRAD256", "0", RESNET), ("NKB_WGRAD256", "2", RESNET), ("NKB_NARROW", "0", RESNET), ("NKB_PLAN", "0", RESNET), ("NKB_PLAN_C", "0", RESNET),
    ("NKB_EVAL_FOLD", "0", RESNET), ("NKB_GEMM8P", "0", UNICOM), ("NKB_FUSED_ATTN", "0", UNICOM), ("NKB_FP8_FUSED_QUANT", "0", UNICOM8),
]
# read at run time but not a choice between two implementations of the step: library path, debugging aids, rehearsal plumbing of
# bench.py / train.py (covered by tests/test_step_semantics_gpu.py), the bucket dtype (tests/test_parallel_gloo.py)
PLUMBING = {"NKBHIP_LIB", "NKB_POISON_WS", "NKB_POISON_LDS", "NKB_CPU_THREADS", "NKB_BENCH_DEVICE", "NKB_DIST_BACKEND", "NKB_FORCE_REDUCER",
            "NKB_DDP_ONE_GPU", "NKB_DDP_BACKEND", "NKB_GRAD_BUCKET_DTYPE"}


def test_switch_surface_is_what_this_file_covers():
    pat = re.compile(r'getenv\("(NKB[A-Z0-9_]*)"\)|environ\.get\("(NKB[A-Z0-9_]*)"|environ\["(NKB[A-Z0-9_]*)"\]|env_int\("(NKB[A-Z0-9_]*)"')
    seen = set()
    files = list((ROOT / "nkb-classification_amd").rglob("*.py")) + list((ROOT / "nkb-classification_amd" / "csrc").glob("*.hip")) + \
        list((ROOT / "nkb-classification_amd" / "csrc").glob("*.h")) + [ROOT / "bench.py"]
    for f in files:
        for m in pat.finditer(f.read_text()):
            seen.add(next(g for g in m.groups() if g))
    seen.discard("NKB_CONVP_DBG")                       # compiled in only with -DNKB_CONVP_STAMPS (scripts/convp_stamps.sh)
    assert len(seen) <= 30, sorted(seen)
    assert seen == {c[0] for c in CASES} | PLUMBING, sorted(seen ^ ({c[0] for c in CASES} | PLUMBING))


@pytest.mark.gpu
@pytest.mark.parametrize("var,value,probe", CASES, ids=[f"{c[0]}={c[1]}" for c in CASES])
def test_train_step_matches_oracle_under_non_default_switch(var, value, probe):
    env = dict(os.environ, **{var: value})
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "switch_probe.py"), *probe], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "PROBE OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
