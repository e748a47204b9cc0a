"""CPU-side checks of the drop-in API: factories, parameter groups, error behaviour, result-dict contract,
state-dict compatibility, and that the HIP product path refuses to run without a GPU (no fallback)."""
import types

import numpy as np
import pytest
import torch

from nkb_classification import losses, metrics, utils
from nkb_classification.logging import BaseLogger
from nkb_classification.model import MultitaskClassifier, SingletaskClassifier, get_model
from oracle.torch_models import OracleClassifier

CFG = dict(model="resnet_tiny_bottleneck", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
           classifier_initialization="kaiming_normal_", task="single")


def test_get_model_surface_and_state_dict_keys():
    m = get_model(CFG, ["a", "b"], "cpu")
    assert isinstance(m, SingletaskClassifier) and m.emb_size == 2048
    o = OracleClassifier(CFG, ["a", "b"])
    assert list(m.state_dict()) == list(o.state_dict())
    m.load_state_dict(o.state_dict())
    assert {"classifier.1.weight", "classifier.1.bias", "emb_model.conv1.weight"} <= set(m.state_dict())
    mm = get_model({**CFG, "task": "multi"}, {"t1": ["a", "b"], "t2": ["x", "y", "z"]}, "cpu")
    assert isinstance(mm, MultitaskClassifier)
    assert {"classifier.t1.1.weight", "classifier.t2.1.bias"} <= set(mm.state_dict())
    m.set_backbone_state("freeze")
    assert not any(p.requires_grad for p in m.emb_model.parameters()) and all(p.requires_grad for p in m.classifier.parameters())
    m.set_backbone_state("unfreeze")
    assert all(p.requires_grad for p in m.emb_model.parameters())
    with pytest.raises(UnboundLocalError):          # model.py:162-177 has no else branch for an unknown task
        get_model({**CFG, "task": "nope"}, ["a"], "cpu")
    with pytest.raises(TypeError, match="nonlinearity"):   # model.py:52-55 behaviour at reference HEAD
        get_model({**CFG, "classifier_initialization": "xavier_normal_"}, ["a", "b"], "cpu")
    with pytest.raises(NotImplementedError):
        get_model({**CFG, "model": "mobilenetv3_large_100"}, ["a"], "cpu")


def test_head_init_matches_reference_rule():
    torch.manual_seed(0)
    m = get_model({**CFG, "model": "resnet50"}, [str(i) for i in range(64)], "cpu")
    w = m.classifier[1].weight
    assert abs(w.std().item() - (2 / 2048) ** 0.5) < 2e-3 and float(m.classifier[1].bias.abs().sum()) == 0.0


def test_no_cpu_fallback():
    m = get_model(CFG, ["a", "b"], "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        losses.get_loss(dict(task="single", type="CrossEntropyLoss"), "cpu")(torch.zeros(2, 3), torch.zeros(2, dtype=torch.long))
    with pytest.raises(RuntimeError, match="only stores parameters"):
        m.emb_model(torch.zeros(1, 3, 64, 64))


def test_get_optimizer_groups_match_reference(golden):
    g2 = golden("g2_optim")
    m = get_model(CFG, ["a", "b", "c"], "cpu")
    for traj in g2["trajectories"]:
        opt = utils.get_optimizer(m, traj["cfg"])
        assert len(opt.param_groups) == 2
        assert len(opt.param_groups[0]["params"]) == len(list(m.emb_model.parameters()))
        assert len(opt.param_groups[1]["params"]) == 2
        for grp, gold in zip(opt.param_groups, g2["group_defaults"][traj["name"]]):
            for k in ("lr", "weight_decay", "betas", "eps", "momentum_decay", "decoupled_weight_decay", "momentum"):
                if k in gold:
                    got = grp[k]
                    assert (list(got) if isinstance(got, tuple) else got) == gold[k], (traj["name"], k)
    with pytest.raises(NotImplementedError, match="Unknown optimizer in config: lion"):
        utils.get_optimizer(m, dict(type="lion"))
    d = utils.get_optimizer(m, dict(type="adam")).param_groups
    assert d[0]["lr"] == 0.001 and d[0]["weight_decay"] == 0.0


def test_get_scheduler_sequences(golden):
    g2 = golden("g2_optim")
    m = get_model(CFG, ["a", "b"], "cpu")
    for name, rec in g2["lr_sequences"].items():
        opt = utils.get_optimizer(m, dict(type="sgd", lr=1.0))
        sch = utils.get_scheduler(opt, rec["policy"])
        seq = []
        for _ in range(5):
            seq.append(opt.param_groups[0]["lr"])
            sch.step()
        assert seq == pytest.approx(rec["lrs"], rel=1e-12, abs=1e-15), name
    assert utils.get_scheduler(utils.get_optimizer(m, dict(type="sgd")), {}) is None
    with pytest.raises(NotImplementedError, match="Learning rate policy poly not implemented."):
        utils.get_scheduler(utils.get_optimizer(m, dict(type="sgd")), dict(type="poly"))


def test_step_scalars_reproduce_torch_coefficients():
    """The host-side scalars fed to nkb_optim_step, checked by replaying torch's formulas on one element."""
    for kind, mk in (("adam", lambda p: torch.optim.Adam([p], lr=1e-2, weight_decay=0.1)),
                     ("nadam", lambda p: torch.optim.NAdam([p], lr=1e-2, weight_decay=0.1, decoupled_weight_decay=True)),
                     ("radam", lambda p: torch.optim.RAdam([p], lr=1e-2, weight_decay=0.1)),
                     ("sgd", lambda p: torch.optim.SGD([p], lr=1e-2, weight_decay=0.1))):
        p = torch.tensor([0.7], dtype=torch.float64, requires_grad=True)
        opt = mk(p)
        w, m, v, state = 0.7, 0.0, 0.0, {}
        for step in range(1, 12):
            g = 0.3 * ((-1) ** step) + 0.05 * step
            p.grad = torch.tensor([g], dtype=torch.float64)
            opt.step()
            k, (c0, c1, c2, _) = utils._step_scalars(kind, state, lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8)
            lr, wd, b1, b2, eps = 1e-2, 0.1, 0.9, 0.999, 1e-8
            if kind == "sgd":
                w = w - lr * (g + wd * w)
            else:
                if kind == "nadam":
                    w = w * (1 - lr * wd)
                else:
                    g = g + wd * w
                m = m + (g - m) * (1 - b1)
                v = v * b2 + (1 - b2) * g * g
                if kind == "adam":
                    w = w - c0 * (m / (v ** 0.5 / c1 + eps))
                elif kind == "nadam":
                    den = (v / c0) ** 0.5 + eps
                    w = w - c1 * (g / den) - c2 * (m / den)
                else:
                    w = w - lr * (m / c0) * c2 * (c1 / (v ** 0.5 + eps)) if c2 > 0 else w - lr * (m / c0)
            assert w == pytest.approx(p.item(), rel=1e-12), (kind, step)


def test_get_loss_factory(golden):
    assert isinstance(losses.get_loss(dict(task="single", type="CrossEntropyLoss", weight=[1, 2]), "cpu"), losses.CrossEntropyLoss)
    f = losses.get_loss(dict(task="single", type="FocalLoss", gamma=1, alpha=[1, 2, .5]), "cpu")
    assert isinstance(f, losses.FocalLoss) and f.gamma == 1 and f.alpha.tolist() == [1, 2, .5]
    assert losses.get_loss(dict(task="single", type="FocalLoss"), "cpu").gamma == losses.DEFAULT_FOCAL_GAMMA == 2.0
    assert isinstance(losses.get_loss(dict(task="multi", type="FocalLoss"), "cpu"), losses.MultitaskCriterion)
    with pytest.raises(NotImplementedError, match=golden("g1_losses")["unknown_type_error"]):
        losses.get_loss(dict(task="single", type="Nope"), "cpu")
    with pytest.raises(ValueError):
        losses.FocalLoss(reduction="avg")


def test_compute_metrics_matches_reference(golden):
    g3 = golden("g3_metrics")
    for name in ("single_C2", "single_C5"):
        got = metrics.compute_metrics(types.SimpleNamespace(task="single"), dict(g3[name]["inputs"]))
        gold = g3[name]["metrics"]
        assert got["epoch_acc"] == pytest.approx(gold["epoch_acc"]) and got["epoch_loss"] == pytest.approx(gold["epoch_loss"])
        np.testing.assert_allclose(got["epoch_roc_auc"], gold["epoch_roc_auc"])
        assert got["loss"] == gold["loss"]
    got = metrics.compute_metrics(types.SimpleNamespace(task="multi", target_names=["a", "b"]), g3["multi"]["inputs"])
    gold = g3["multi"]["metrics"]
    assert got["epoch_acc"] == pytest.approx(gold["epoch_acc"]) and got["loss"] == gold["loss"]
    for t in ("a", "b"):
        assert got[t]["epoch_acc"] == pytest.approx(gold[t]["epoch_acc"])
        np.testing.assert_allclose(got[t]["epoch_roc_auc"], gold[t]["epoch_roc_auc"])
    with pytest.raises(ValueError):
        metrics.compute_metrics(types.SimpleNamespace(task="x"), {})


def test_logger_construction_and_contract():
    lg = BaseLogger(types.SimpleNamespace(task="multi"), {"b": ["x"], "a": ["y", "z"]})
    assert lg.target_names == ["a", "b"]            # logging.py:243 crashes here at reference HEAD; intent = sorted names
    res = lg.get_epoch_results()
    assert set(res) == {"running_loss", "confidences", "predictions", "ground_truth", "images"}
    with pytest.raises(AssertionError):
        BaseLogger(types.SimpleNamespace(task="other"), [])


def test_read_py_config(tmp_path):
    p = tmp_path / "my_cfg.py"
    p.write_text("device = 'cuda:0'\n")
    line = utils.read_py_config(str(p))
    assert line == "import my_cfg as cfg"
    ns = {}
    exec(line, ns, ns)
    assert ns["cfg"].device == "cuda:0"
    assert utils.get_classes_configs(["a", "b"]) == ({"a": 0, "b": 1}, {0: "a", 1: "b"})


@pytest.mark.parametrize("backbone,task", [("resnet_tiny_basic", "single"), ("resnet_tiny_bottleneck", "multi"),
                                           ("vit_tiny_test", "single"), ("unicom ViT-tiny-test", "single")])
def test_scripted_export_matches_oracle(tmp_path, backbone, task):
    """train.py:66-73 counterpart: the TorchScript archive written next to each checkpoint carries the trained weights
    and the reference's forward contract (Tensor / dict of Tensors) — checked against the oracle on the same state."""
    from nkb_classification.model import get_model
    from nkb_classification.scripted import save_scripted
    from oracle.torch_models import OracleClassifier
    cfg_model = dict(model=backbone, pretrained=False, backbone_dropout=0.1, classifier_dropout=0.1,
                     classifier_initialization="kaiming_normal_", task=task)
    classes = ["a", "b", "c"] if task == "single" else {"t1": ["x", "y"], "t2": ["p", "q", "r"]}
    torch.manual_seed(0)
    oracle = OracleClassifier(cfg_model, classes).eval()
    model = get_model(dict(cfg_model), classes, "cpu")
    model.load_state_dict(oracle.state_dict())
    path = tmp_path / "scripted_last.pt"
    save_scripted(model, path)
    loaded = torch.jit.load(str(path))
    hw = 56 if backbone.startswith("unicom") else 64
    x = torch.randn(2, 3, hw, hw)
    with torch.no_grad():
        ref, out = oracle(x), loaded(x)
    if task == "single":
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    else:
        assert sorted(out) == sorted(ref)
        for t in ref:
            torch.testing.assert_close(out[t], ref[t], rtol=1e-5, atol=1e-5)


def test_unicom_family_matches_reference_call_site():
    """model.py:75-79: `unicom.load(name.split()[1])[0]`, emb_size = feature[-2].out_features; key names and shapes of the
    product containers equal the oracle restatement's (so unicom checkpoints load into either)."""
    from nkb_classification.model import SingletaskClassifier
    from nkb_classification import unicom
    from oracle.torch_models import create_backbone
    with torch.device("meta"):
        prod, orc = unicom.load("ViT-L/14"), create_backbone("unicom ViT-L/14")
    ps, os_ = prod.state_dict(), orc.state_dict()
    assert list(ps) == list(os_) and all(ps[k].shape == os_[k].shape for k in ps)
    assert sum(p.numel() for p in prod.parameters()) == 572_328_448
    emb_model, emb_size = SingletaskClassifier.get_emb_model(dict(model="unicom ViT-tiny-test", pretrained=False))
    assert emb_size == emb_model.feature[-2].out_features == 64
    with pytest.raises(RuntimeError, match="not found"):
        unicom.load("ViT-H/14")


def test_samplers_match_reference_formulas():
    """dataset.py:27-86 (weights 1 / count[label], multinomial with replacement from the global generator) and the
    data-parallel split of SURVEY.md §8(e) (seed-shared permutation, rank r takes r::W)."""
    from nkb_classification.dataset import ImbalancedDatasetSampler, ShardedSampler

    class DS(torch.utils.data.Dataset):
        labels = [0, 0, 0, 0, 0, 0, 0, 1, 1, 2]

        def __len__(self):
            return len(self.labels)

        def get_labels(self):
            return self.labels

    s = ImbalancedDatasetSampler(DS())
    assert len(s) == 10
    torch.testing.assert_close(s.weights, torch.tensor([1 / 7] * 7 + [0.5, 0.5, 1.0], dtype=torch.float64))
    torch.manual_seed(4)
    drawn = list(s)
    torch.manual_seed(4)
    assert drawn == torch.multinomial(s.weights, 10, replacement=True).tolist()
    sub = ImbalancedDatasetSampler(DS(), indices=[0, 7, 9], num_samples=50)
    assert len(sub) == 50 and set(sub) <= {0, 7, 9} and sub.weights.tolist() == [1.0, 1.0, 1.0]
    # per-rank streams (seed + rank), each drawing its share
    r0, r1 = (ImbalancedDatasetSampler(DS(), seed=3, rank=r, world=2) for r in (0, 1))
    assert len(r0) == len(r1) == 5 and list(r0) != list(r1)
    shards = [ShardedSampler(10, r, 4, shuffle=True, seed=3) for r in range(4)]
    got = [list(sh) for sh in shards]
    assert all(len(g) == 3 for g in got) and set(sum(got, [])) == set(range(10))
    perm = torch.randperm(10, generator=torch.Generator().manual_seed(3)).tolist()
    assert got[1] == (perm + perm[:2])[1::4]
    for sh in shards:
        sh.set_epoch(1)
    assert [list(sh) for sh in shards] != got
    assert list(ShardedSampler(5, 1, 2, shuffle=False)) == [1, 3, 0]


def test_get_dataset_sampling_options(tmp_path):
    """get_dataset (dataset.py:541-629): weighted_sampling / shuffle / drop_last / world sharding reach the DataLoader."""
    from PIL import Image
    from nkb_classification.dataset import ImbalancedDatasetSampler, ShardedSampler, get_dataset
    import numpy as np
    for cls, n in (("a", 5), ("b", 2)):
        (tmp_path / cls).mkdir()
        for i in range(n):
            Image.fromarray(np.full((12 + i, 20, 3), 40 * i, np.uint8)).save(tmp_path / cls / f"{i}.png")
    base = dict(root=str(tmp_path), size=16, batch_size=2, num_workers=0)
    ld = get_dataset(dict(base, weighted_sampling=True, drop_last=True))
    assert isinstance(ld.sampler, ImbalancedDatasetSampler) and ld.drop_last and ld.dataset.classes == ["a", "b"]
    assert ld.sampler.weights.tolist() == [0.2] * 5 + [0.5] * 2
    ld = get_dataset(dict(base, shuffle=True, rank=1, world=2, seed=5))
    assert isinstance(ld.sampler, ShardedSampler) and len(ld.sampler) == 4
    x, y = next(iter(get_dataset(dict(base, shuffle=False))))
    assert x.shape == (2, 3, 16, 16) and x.dtype == torch.float32 and y.tolist() == [0, 0]
    # raw mode of the folder source (what DeviceLoader consumes): longest side scaled to `size`, top-left placement
    from nkb_classification.dataset import FolderDataset
    raw, hw, label = FolderDataset(str(tmp_path), size=16, raw=True)[0]
    assert raw.dtype == torch.uint8 and raw.shape == (16, 16, 3) and hw.tolist() == [10, 16] and label == 0
    assert raw[10:].abs().sum() == 0


def test_changing_the_reserved_cus_invalidates_recorded_plans():
    """nkb_rowres_reserve_cus sizes the grids (partial-sum rows, slab counts) of the backward kernels; a launch plan recorded
    under another value must not be replayed.  Workspace.generation — what a plan is keyed on — carries the setting's epoch."""
    from nkb_classification import hip, runtime
    ws = runtime.Workspace("cpu")
    try:
        hip.rowres_reserve_cus(0)
        g0 = ws.generation
        hip.rowres_reserve_cus(0)                       # same value: plans stay
        assert ws.generation == g0
        hip.rowres_reserve_cus(32)
        g1 = ws.generation
        assert g1 != g0 and hip.load().nkb_rowres_reserved_cus() == 32
        ws.get("a", (4,), __import__("torch").float32)  # allocations still count
        assert ws.generation != g1
    finally:
        hip.rowres_reserve_cus(0)
