"""Pins the CPU oracle (oracle/) against the golden vectors captured by importing the reference's own modules
(oracle/make_golden.py -> tests/golden/*.json).  Runs on CPU; tolerances allow for a different host CPU's
vectorisation (the goldens were generated in the build container)."""
import math
import types

import numpy as np
import pytest
import torch

from oracle import torch_engine as oe
from oracle.torch_models import OracleClassifier, count_params, create_backbone


def test_g1_losses(golden):
    g1 = golden("g1_losses")
    assert "Unknown loss type in config: Nope" == g1["unknown_type_error"]
    with pytest.raises(NotImplementedError, match="Unknown loss type in config: Nope"):
        oe.Criterion(dict(task="single", type="Nope"))
    for case in g1["cases"]:
        crit = oe.Criterion(case["cfg"])
        if case["cfg"]["task"] == "multi":
            xs = {k: torch.tensor(v, requires_grad=True) for k, v in case["x"].items()}
            ys = {k: torch.tensor(v) for k, v in case["y"].items()}
            out = crit(xs, ys)
            out["loss"].backward()
            for k, v in case["loss"].items():
                assert out[k].item() == pytest.approx(v, rel=1e-6, abs=1e-7), (case["name"], k)
            for k, v in case["grad"].items():
                torch.testing.assert_close(xs[k].grad, torch.tensor(v), rtol=1e-5, atol=1e-7)
        else:
            x = torch.tensor(case["x"], requires_grad=True)
            y = torch.tensor(case["y"])
            out = crit(x, y)
            assert out.item() == pytest.approx(case["loss"], rel=1e-6, abs=1e-7), case["name"]
            if case["grad"] is not None:
                out.backward()
                torch.testing.assert_close(x.grad, torch.tensor(case["grad"]), rtol=1e-5, atol=1e-7, msg=case["name"])
    named = {c["name"]: c["loss"] for c in g1["cases"]}
    # the values SURVEY.md §8(c) recorded from its own probe of the reference
    assert named["ce"] == pytest.approx(0.5902279019, rel=1e-6)
    assert named["ce_weighted"] == pytest.approx(0.5413813591, rel=1e-6)
    assert named["focal_g2"] == pytest.approx(0.2251573503, rel=1e-6)
    assert named["focal_g1_alpha"] == pytest.approx(0.4585762322, rel=1e-6)
    assert named["focal_all_ignored"] == 0.0
    assert named["multi_focal_g1"]["loss"] == pytest.approx(0.5183875561, rel=1e-6)


class _Tiny(torch.nn.Module):
    def __init__(self, init):
        super().__init__()
        self.emb_model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4))
        self.classifier = torch.nn.Sequential(torch.nn.Dropout(0.0), torch.nn.Linear(4, 3))
        self.load_state_dict({k: torch.tensor(v) for k, v in init.items()})

    def forward(self, x):
        return self.classifier(self.emb_model(x))


def test_g2_optimizer_trajectories_and_schedules(golden):
    g2 = golden("g2_optim")
    for traj in g2["trajectories"]:
        m = _Tiny(traj["init"])
        opt = oe.make_optimizer(m, traj["cfg"])
        for grp, gold in zip(opt.param_groups, g2["group_defaults"][traj["name"]]):
            for k, v in gold.items():
                got = grp[k]
                assert (list(got) if isinstance(got, tuple) else got) == v, (traj["name"], k)
        for x, y, step in zip(traj["x"], traj["y"], traj["steps"]):
            opt.zero_grad()
            torch.nn.functional.cross_entropy(m(torch.tensor(x)), torch.tensor(y)).backward()
            opt.step()
            for k, v in step["params"].items():
                torch.testing.assert_close(m.state_dict()[k], torch.tensor(v), rtol=1e-5, atol=1e-7, msg=f"{traj['name']} {k}")
    nadam = g2["group_defaults"]["nadam"][0]
    assert nadam["decoupled_weight_decay"] is True and nadam["momentum_decay"] == 0.004
    assert g2["group_defaults"]["sgd"][0]["momentum"] == 0
    for name, rec in g2["lr_sequences"].items():
        m = _Tiny(g2["trajectories"][0]["init"])
        opt = oe.make_optimizer(m, dict(type="sgd", lr=1.0))
        sch = oe.make_scheduler(opt, rec["policy"])
        seq = []
        for _ in range(5):
            seq.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        assert seq == pytest.approx(rec["lrs"], rel=1e-12, abs=1e-15), name
    assert g2["lr_sequences"]["step"]["lrs"] == pytest.approx([1, 1, .5, .5, .25])
    assert g2["lr_sequences"]["multistep"]["lrs"] == pytest.approx([1, .1, .1, .01, .01])
    assert g2["lr_sequences"]["cosine"]["lrs"] == pytest.approx([1, .853553, .5, .146447, 0], abs=1e-6)
    assert g2["empty_policy_is_none"] is True and oe.make_scheduler(None, {}) is None
    assert g2["unknown_optimizer_error"] == "Unknown optimizer in config: lion"
    assert g2["unknown_scheduler_error"] == "Learning rate policy poly not implemented."


def _engine_case(case):
    classes = case["classes"]
    multi = isinstance(classes, dict)
    torch.manual_seed(case["seed"])
    model = OracleClassifier(case["cfg_model"], classes)
    n_cls = {t: len(c) for t, c in classes.items()} if multi else len(classes)
    train = oe.synthetic_batches(case["n_images"], case["batch"], n_cls, seed=1234, hw=case["hw"])
    val = oe.synthetic_batches(2 * case["batch"], case["batch"], n_cls, seed=4321, hw=case["hw"])
    opt = oe.make_optimizer(model, case["opt_cfg"])
    sch = oe.make_scheduler(opt, dict(type="cosine", n_epochs=case["n_epochs_cos"]))
    crit = oe.Criterion(case["crit_cfg"])
    log = oe.EpochLog(multi)
    for gold in case["epochs"]:
        tr = oe.train_epoch(model, train, opt, sch, crit, log, log_gradients="grad_total" in gold)
        tr = dict(tr)
        tr_loss = tr["running_loss"]
        lr_after = [g["lr"] for g in opt.param_groups]
        grad_total = [float(v) for v in tr["metrics_grad_log"]["Gradients/Total"]] if "grad_total" in gold else None
        va = oe.val_epoch(model, val, crit, log)
        if multi:
            for k, v in gold["train_running_loss"].items():
                assert tr_loss[k] == pytest.approx(v, rel=2e-4, abs=1e-6), k
            for k, v in gold["val_running_loss"].items():
                assert va["running_loss"][k] == pytest.approx(v, rel=2e-4, abs=1e-6), k
            for k, v in gold["val_predictions"].items():
                assert va["predictions"][k] == v
        else:
            assert tr_loss == pytest.approx(gold["train_running_loss"], rel=2e-4, abs=1e-6)
            assert va["running_loss"] == pytest.approx(gold["val_running_loss"], rel=2e-4, abs=1e-6)
            assert va["predictions"] == gold["val_predictions"]
            np.testing.assert_allclose(va["confidences"], gold["val_confidences"], rtol=2e-4, atol=1e-6)
        assert lr_after == pytest.approx(gold["lr_after"], rel=1e-12)
        if grad_total is not None:
            assert grad_total == pytest.approx(gold["grad_total"], rel=1e-3)
            assert gold["grad_keys"][0].startswith("Gradients/")
    for k, v in case["param_norms"].items():
        assert float(model.state_dict()[k].float().norm()) == pytest.approx(v, rel=2e-4, abs=1e-6), k


@pytest.mark.parametrize("name", ["tiny_basic_single", "tiny_bottleneck_multi", "tiny_vit_single"])
def test_g4_engine_trajectory_small(golden, name):
    _engine_case(golden("g4_engine")[name])


def test_g4_engine_trajectory_config1_resnet18(golden):
    """BASELINE config 1 on the oracle: per-iteration losses, cosine lr after one of five epochs, grad norms."""
    case = golden("g4_engine")["config1_resnet18"]
    _engine_case(case)
    assert case["epochs"][0]["lr_after"][0] == pytest.approx(0.90450849718 * 1e-4, rel=1e-9)


def test_g5_logger_lists(golden):
    """Lists produced by the REFERENCE's BaseLogger.log_iter / get_epoch_results (logging.py:245-294, captured by
    oracle/make_golden.py with stand-in modules for its unrelated imports) vs the oracle's epoch log."""
    g5 = golden("g5_logger")
    probe = g5["probe"]
    log = oe.EpochLog(False)
    log.add(torch.tensor(probe["preds"]), torch.tensor(probe["true"]), torch.tensor(probe["loss"]))
    res = log.results()
    np.testing.assert_allclose(res["confidences"], probe["confidences"], rtol=1e-6)
    assert res["predictions"] == probe["predictions"] == [1, 0] and res["ground_truth"] == probe["ground_truth"]
    assert res["running_loss"] == probe["running_loss"] == [0.75]
    assert set(res) == {"running_loss", "confidences", "predictions", "ground_truth", "images"}
    single = g5["single"]
    log = oe.EpochLog(False)
    for b in single["batches"]:
        log.add(torch.tensor(b["preds"]), torch.tensor(b["true"]), torch.tensor(b["loss"]))
    res = log.results()
    np.testing.assert_allclose(res["confidences"], single["confidences"], rtol=1e-6)
    np.testing.assert_allclose(res["running_loss"], single["running_loss"], rtol=1e-7)
    assert res["predictions"] == single["predictions"] and res["ground_truth"] == single["ground_truth"]
    multi = g5["multi"]
    log = oe.EpochLog(True)
    for b in multi["batches"]:
        log.add({t: torch.tensor(v) for t, v in b["preds"].items()}, {t: torch.tensor(v) for t, v in b["true"].items()},
                {t: torch.tensor(v) for t, v in b["loss"].items()})
    res = log.results()
    assert set(res["running_loss"]) == set(multi["running_loss"]) == {"shape", "color", "loss"}
    for t in multi["classes"]:
        np.testing.assert_allclose(res["confidences"][t], multi["confidences"][t], rtol=1e-6)
        assert res["predictions"][t] == multi["predictions"][t] and res["ground_truth"][t] == multi["ground_truth"][t]
    for t in multi["running_loss"]:
        np.testing.assert_allclose(res["running_loss"][t], multi["running_loss"][t], rtol=1e-7)


def test_backbone_invariants():
    """timm topology restated from memory: pinned by parameter counts, feature widths and key names."""
    r18, r50 = create_backbone("resnet18"), create_backbone("resnet50")
    assert count_params(r18) == 11_176_512 and r18.num_features == 512
    assert count_params(r50) == 23_508_032 and r50.num_features == 2048
    vit = create_backbone("vit_base_patch16_224")
    assert count_params(vit) == 85_798_656 and vit.num_features == 768
    keys = set(r50.state_dict())
    for k in ("conv1.weight", "bn1.running_var", "layer1.0.downsample.0.weight", "layer1.0.downsample.1.bias",
              "layer4.2.conv3.weight", "layer3.5.bn2.num_batches_tracked"):
        assert k in keys
    vk = set(vit.state_dict())
    for k in ("cls_token", "pos_embed", "patch_embed.proj.weight", "blocks.0.attn.qkv.bias", "blocks.11.mlp.fc2.weight",
              "norm.bias"):
        assert k in vk
    # unicom family (model.py:77-79) — restated from SURVEY §8 A9, parity unpinned: 572.33 M parameters for ViT-L/14
    # (268.4 M of them in feature.0), 768-wide feature[-2], 256 tokens, bias-free qkv
    with torch.device("meta"):
        uni = create_backbone("unicom ViT-L/14")
    assert count_params(uni) == 572_328_448 and uni.feature[-2].out_features == 768 == uni.num_features
    assert uni.feature[0].weight.shape == (1024, 256 * 1024) and uni.pos_embed.shape == (1, 256, 1024)
    uk = set(uni.state_dict())
    for k in ("pos_embed", "patch_embed.proj.bias", "blocks.23.attn.qkv.weight", "blocks.0.mlp.fc1.bias", "norm.weight",
              "feature.0.weight", "feature.1.running_var", "feature.3.num_batches_tracked"):
        assert k in uk
    assert "blocks.0.attn.qkv.bias" not in uk and "cls_token" not in uk and "feature.0.bias" not in uk
    with torch.no_grad():
        assert create_backbone("unicom ViT-tiny-test").eval()(torch.zeros(2, 3, 56, 56)).shape == (2, 64)
    with torch.no_grad():
        assert r18.eval()(torch.zeros(1, 3, 64, 64)).shape == (1, 512)
        assert create_backbone("vit_tiny_test").eval()(torch.zeros(2, 3, 64, 64)).shape == (2, 128)
    # MaxPool tie rule the HIP kernel must reproduce: first element in row-major window order wins
    x = torch.zeros(1, 1, 4, 4, requires_grad=True)
    torch.nn.functional.max_pool2d(x, 3, 2, 1).sum().backward()
    assert x.grad.flatten().nonzero().flatten().tolist() == [0, 1, 4, 5]
