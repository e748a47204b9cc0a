"""Test harness around nkb-classification_amd/train.py (tests/test_step_semantics_gpu.py launches THIS file under
torch.distributed.run): runs train.main() unchanged and afterwards writes what this rank ended up with — flat parameters, buffers,
the classifier weight, how many samples the gathered epoch results hold — to $NKB_DUMP_PARAMS/params_rank<r>.pt."""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "nkb-classification_amd"))
import train  # noqa: E402


def _dump(model, train_loader, val_loader, results, cfg):
    from nkb_classification import parallel
    out = os.environ["NKB_DUMP_PARAMS"]
    train_results, val_results = results

    def count(res):
        if res is None:
            return -1
        gt = res["ground_truth"]
        return len(gt) if isinstance(gt, list) else len(next(iter(gt.values())))
    sampler = getattr(getattr(train_loader, "loader", train_loader), "sampler", None)
    head = [v for k, v in model.state_dict().items() if k.startswith("classifier") and k.endswith("weight")]
    torch.save(dict(flat_param=model.arena.flat_param.cpu(), buffers={k: v.cpu() for k, v in model.named_buffers()},
                    head_weight=head[0].cpu() if head else None, n_train=count(train_results), n_val=count(val_results),
                    dataset_len=len(train_loader.dataset), val_len=len(val_loader.dataset),
                    shard_len=len(sampler) if sampler is not None else -1),
               Path(out, f"params_rank{parallel.rank()}.pt"))


if __name__ == "__main__":
    train._after_train = _dump
    train.main()
