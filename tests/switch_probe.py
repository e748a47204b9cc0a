"""Child process of tests/test_switches_gpu.py: one model, a few train steps next to the oracle (tests/test_parity_bench_size_gpu._run),
under whatever NKB_* values the parent put into the environment — the switches are read once per process (module constants, function
statics in libnkbhip), so every value needs a process of its own.  Prints one line `PROBE OK {...}` or raises."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "nkb-classification_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
os.environ.setdefault("MIOPEN_FIND_MODE", "2")

import torch  # noqa: E402

import test_parity_bench_size_gpu as pb  # noqa: E402


def main():
    model, batch, dtype, size = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    out, counters, plans = pb._run(model, batch, dtype, 1000, steps=3 if dtype != "fp8" else 6, size=size)
    if dtype == "fp8":
        pb._check(out, relative=False, cos_bar=0.93, l2_bar=0.36, loss_tol=2e-2)
    else:
        pb._check(out, relative=True, cos_slack=2e-3)
    if os.environ.get("NKB_EVAL_FOLD") is not None:
        # the evaluation path of the same model next to the oracle in eval mode (running statistics after the steps above differ
        # between the two only by the bf16 step: compare on the HIP model's own statistics)
        import argparse
        import bench
        from oracle.torch_models import OracleClassifier
        m, _, _ = bench.build(argparse.Namespace(model=model, classes=1000, batch=batch, dtype=dtype, heads=""), torch.device(pb.DEV))
        o = OracleClassifier(dict(task="single", model=model, pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                                  classifier_initialization="kaiming_normal_"), [str(i) for i in range(1000)])
        o.load_state_dict(m.state_dict())
        o = o.to(pb.DEV).eval()
        m.eval()
        x = torch.randn(16, 3, size, size, generator=torch.Generator().manual_seed(5)).to(pb.DEV)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            a = m(x).float()
        with torch.no_grad():
            b = o(x).float()
        err = ((a - b).abs().max() / b.abs().max()).item()
        assert err < 3e-2, err
    print("PROBE OK " + json.dumps(dict(counters=counters, plans=plans, last=out[-1])))


if __name__ == "__main__":
    main()
