"""Classifier wrappers and `get_model` — drop-in for /root/reference/nkb_classification/model.py.

`SingletaskClassifier` / `MultitaskClassifier` keep the reference's constructor arguments, attributes
(`emb_model`, `emb_size`, `classifier`), methods (`set_backbone_state`, `set_dropout`,
`initialize_classifier[s]`) and state-dict key names (model.py:17-159).  `forward` does not run torch
modules: the whole network (backbone + fused heads) is ONE autograd node whose forward and backward enqueue
libnkbhip kernels on the current HIP stream and whose parameter gradients land in the flat gradient arena.

Compute dtype: fp32 (exact-fp32 MFMA; the parity mode, equal to the reference's CPU path) unless the call
happens under `torch.autocast` (what engine.train_epoch does when cfg.enable_mixed_presicion is set), in
which case activations / GEMM operands are bf16 with fp32 accumulation, statistics, logits and master weights.
"""
from __future__ import annotations

import os
from typing import Dict, List, Union

import torch
from torch import nn

from . import hip
from .backbones import create_backbone
from .hipnet import HipEngine
from .runtime import ParamArena


# recorded launch plans for the train step (hip.Plan): 0 = every step goes through the Python layer code
_PLANS = os.environ.get("NKB_PLAN", "1") != "0"


def _alloc_count(device) -> int:
    """Number of device allocations torch has served so far (cache hits included), without those made inside hip.host_op closures
    (the DDP bucket hooks): a closure is re-executed at every replay and allocates its temporaries afresh, so they never end up as
    frozen addresses in a plan."""
    return int(torch.cuda.memory_stats(device).get("allocation.all.allocated", 0)) - hip.host_op_allocs


_warned_plan_rejected = set()


def _plan_rejected(kind: str, key):
    """A recorded plan was thrown away because the recorded run allocated device memory: say so once per plan key — silently the step
    would stay on the Python path and re-record forever."""
    if key not in _warned_plan_rejected:
        _warned_plan_rejected.add(key)
        import warnings
        warnings.warn(f"nkb_classification: the recorded {kind} launch plan was discarded (the recorded step allocated device memory "
                      "outside the workspace); this configuration keeps running through the Python layer code", RuntimeWarning)


class _NetFn(torch.autograd.Function):
    """model(img) as a single autograd node.  Parameters are passed only so autograd schedules backward; their
    gradients are written straight into the arena (returned as None here) — see ParamArena.publish_grads."""

    @staticmethod
    def forward(ctx, owner, img, *params):
        ctx.owner = owner
        ctx.token = owner._fwd_token
        return owner._forward_impl(img)

    @staticmethod
    def backward(ctx, glogits):
        owner = ctx.owner
        if ctx.token != owner._fwd_token:
            raise RuntimeError("HIP engine: backward() called after a newer forward overwrote the saved "
                               "activations (only the most recent forward can be differentiated)")
        with torch.cuda.device(glogits.device):
            owner._backward_impl(glogits)
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _HipClassifier(nn.Module):
    def __init__(self, cfg_model: dict):
        super().__init__()
        self.emb_model, self.emb_size = self.get_emb_model(cfg_model)
        self.set_dropout(self.emb_model, cfg_model["backbone_dropout"])
        self.arena = ParamArena()
        self._engines: Dict[torch.dtype, HipEngine] = {}
        self._fwd_token = 0
        self._active: HipEngine = None
        self._nbt_flat = None
        self.grad_ready_hook = None   # set by parallel.GradReducer: called as hook(lo, hi) while backward runs
        self.grad_done_hook = None    # set by parallel.GradReducer: called once at the end of backward
        # fp8 Linear contractions under bf16 autocast (BASELINE configs[4]; cfg.amp_dtype = "fp8"): e4m3 activations and
        # weights forward, e5m2 gradients x e4m3 weights for the data gradients, bf16 weight gradients, fp32 masters
        self.fp8_linear = False

    # ---- reference API -------------------------------------------------------------------
    @staticmethod
    def get_emb_model(cfg_model: dict):
        name = cfg_model["model"]
        if name.lower().startswith("unicom"):
            # model.py:77-79: `unicom.load(name.split()[1])[0]`, embedding width = feature[-2].out_features
            from . import unicom
            emb_model = unicom.load(name.split()[1])
            return emb_model, emb_model.feature[-2].out_features
        emb_model = create_backbone(name, pretrained=cfg_model.get("pretrained", False))
        return emb_model, emb_model.num_features

    @staticmethod
    def set_dropout(model: nn.Module, drop_rate: float = 0.2) -> None:
        for child in model.children():
            if isinstance(child, nn.Dropout):
                child.p = drop_rate
            _HipClassifier.set_dropout(child, drop_rate=drop_rate)

    def set_backbone_state(self, state: str):
        for p in self.emb_model.parameters():
            if state == "freeze":
                p.requires_grad = False
            elif state == "unfreeze":
                p.requires_grad = True

    @staticmethod
    def _init_params(params, strategy: str):
        for p in params:
            if p.ndim >= 2:
                if strategy == "kaiming_normal_":
                    nn.init.kaiming_normal_(p, nonlinearity="relu")
                elif strategy == "kaiming_uniform_":
                    nn.init.kaiming_uniform_(p, nonlinearity="relu")
                elif strategy in ("xavier_normal_", "xavier_uniform_"):
                    # model.py:52-55 passes nonlinearity= to xavier_*, which torch rejects with this TypeError
                    raise TypeError(f"{strategy}() got an unexpected keyword argument 'nonlinearity'")
            else:
                nn.init.zeros_(p)

    # ---- packing / engines ------------------------------------------------------------------
    def _heads(self) -> List[nn.Linear]:
        raise NotImplementedError

    def _pack(self, device):
        heads = self._heads()
        blocks = [[p] for p in self.emb_model.parameters()]
        blocks += [[h.weight] for h in heads]
        blocks.append([h.bias for h in heads])
        self.arena.pack(blocks, device)
        # one int64 word per BatchNorm for num_batches_tracked, bumped by a single add per step
        bns = [m for m in self.emb_model.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        if bns:
            flat = torch.stack([m.num_batches_tracked.to(device).reshape(()) for m in bns]).contiguous()
            for i, m in enumerate(bns):
                m._buffers["num_batches_tracked"] = flat[i]
            self._nbt_flat = flat
        self._engines.clear()

    def _engine(self, device, dtype) -> HipEngine:
        if not self.arena.still_packed() or self.arena.device != device:
            self._pack(device)
        eng = self._engines.get(dtype)
        if eng is None:
            eng = HipEngine(self.arena, device, dtype)
            heads = self._heads()
            em = self.emb_model
            eng.register(em.gemm_convs(), em.stem_convs(), [h.weight for h in heads], [h.bias for h in heads])
            if hasattr(em, "fp8_linears"):
                eng.fp8_candidates = em.fp8_linears()
            self._engines[dtype] = eng
        return eng

    def _apply(self, fn, *a, **k):
        # module.to()/cuda()/float() re-create parameter storage: drop the arena, it is rebuilt on the next forward
        out = super()._apply(fn, *a, **k)
        self.arena.packed = False
        return out

    def _load_from_state_dict(self, *a, **k):
        out = super()._load_from_state_dict(*a, **k)
        self.arena.mark_dirty()
        return out

    # ---- execution -----------------------------------------------------------------------------
    def _compute_dtype(self) -> torch.dtype:
        return torch.bfloat16 if torch.is_autocast_enabled() else torch.float32

    def _classifier_dropout_p(self) -> float:
        raise NotImplementedError

    def train(self, mode: bool = True):
        # every train()/eval() switch starts a new eval phase: folded eval-mode filters are rebuilt on first use
        self._eval_phase = getattr(self, "_eval_phase", 0) + 1
        return super().train(mode)

    def _forward_impl(self, img: torch.Tensor) -> torch.Tensor:
        eng = self._active
        train = self.training
        need_dgrad = train and any(p.requires_grad for p in self.emb_model.parameters())
        if eng.fp8 != bool(self.fp8_linear and train):
            eng.fp8 = bool(self.fp8_linear and train)
            eng._wver = -1                          # the fp8 weight copies follow the switch
        eng.refresh_weights(need_dgrad=need_dgrad)
        eng.fp8_begin_step()
        eng.fold_key = (self.arena.version, getattr(self, "_eval_phase", 0))
        drop_p = self._classifier_dropout_p() if train else 0.0
        ctot = sum(h.out_features for h in self._heads())
        logits = torch.empty(img.shape[0], ctot, device=img.device, dtype=torch.float32)

        def run():
            emb = self.emb_model.run_forward(eng, img, train)
            if train and self._nbt_flat is not None:
                hip.host_op(lambda: self._nbt_flat.add_(1))
            eng.head(emb, train, drop_p, out=logits)

        # the backward pass of this forward: which plan it may use (None = the Python path)
        self._fwd_key = None
        if not (_PLANS and train):
            run()
            return logits
        # (the arena's base address and the requires_grad pattern are part of the key: a re-packed arena or a changed freeze
        # pattern must never replay pointers / launch sequences recorded for the old one — ADVICE r2)
        key = ("fwd", tuple(img.shape), need_dgrad, drop_p, eng.overlap_wgrad, self.arena.total, eng.fp8, eng.gram_bn,
               self.arena.flat_param.data_ptr(), hash(tuple(p.requires_grad for p in self.parameters())))
        self._fwd_key = key
        ent = eng.plans.get(key)
        if ent is not None and ent[1] == eng.ws.generation:
            eng.saved = dict(ent[2])               # the activation table this plan (and its backward twin) was recorded with
            hip.replay(ent[0], {"img": img, "logits": logits})
            return logits
        # record on the second consecutive run that needed no new buffers (the first one sizes the workspace)
        record = eng.plan_seen.get(key) == eng.ws.generation
        if record:
            hip.record_begin()
        allocs = _alloc_count(img.device) if record else 0
        try:
            run()
        except BaseException:
            hip.record_abort()
            raise
        if record:
            plan = hip.record_end({"img": img, "logits": logits})
            # a plan freezes raw addresses: it is only kept when the recorded run allocated NOTHING (every buffer it touched is a
            # persistent workspace / arena / engine buffer) — a torch temporary created inside run() would be replayed after its free
            if eng.plan_seen.get(key) == eng.ws.generation and _alloc_count(img.device) == allocs:
                eng.plans[key] = (plan, eng.ws.generation, dict(eng.saved))
            elif eng.plan_seen.get(key) == eng.ws.generation:
                _plan_rejected("forward", key)
        eng.plan_seen[key] = eng.ws.generation
        return logits

    def _backward_impl(self, glogits: torch.Tensor):
        eng = self._active
        arena = self.arena
        heads = self._heads()
        head_params = [h.weight for h in heads] + [h.bias for h in heads]
        bb_params = [p for p in self.emb_model.parameters() if p.requires_grad]
        wanted = [p for p in head_params if p.requires_grad] + bb_params
        arena.begin_backward(wanted)
        if not glogits.is_contiguous():
            glogits = glogits.contiguous()
        hook = self.grad_ready_hook

        def run():
            g_emb = eng.head_backward(glogits, need_demb=bool(bb_params))
            if hook is not None:
                lo = arena.offset_of(heads[0].weight)
                hip.host_op(lambda: hook(lo, arena.total, side_event=eng.side_event()))
            if bb_params:
                on_done = None
                if hook is not None:
                    def on_done(module):
                        # the range is final once the compute stream AND the weight-gradient stream reach this point; the
                        # communication stream waits for both, the compute stream for neither
                        rng = arena.range_of(list(module.parameters()) if isinstance(module, nn.Module) else list(module))
                        if rng is not None:
                            hip.host_op(lambda: hook(*rng, side_event=eng.side_event()))
                self.emb_model.run_backward(eng, g_emb, on_done)
            eng.wait_side()

        fwd_key = getattr(self, "_fwd_key", None)
        if fwd_key is None or fwd_key not in eng.plans or eng.plans[fwd_key][1] != eng.ws.generation:
            run()                                  # no forward plan (yet): the activation table may still change
            if fwd_key is not None:
                eng.plan_seen.pop(("bwd", fwd_key), None)
        else:
            key = ("bwd", fwd_key, len(wanted), len(bb_params), hook is not None)
            ent = eng.plans.get(key)
            if ent is not None and ent[1] == eng.ws.generation:
                hip.replay(ent[0], {"glogits": glogits})
            else:
                record = eng.plan_seen.get(key) == eng.ws.generation
                if record:
                    hip.record_begin()
                allocs = _alloc_count(glogits.device) if record else 0
                try:
                    run()
                except BaseException:
                    hip.record_abort()
                    raise
                if record:
                    plan = hip.record_end({"glogits": glogits})
                    if eng.plan_seen.get(key) == eng.ws.generation and _alloc_count(glogits.device) == allocs:
                        eng.plans[key] = (plan, eng.ws.generation, None)
                    elif eng.plan_seen.get(key) == eng.ws.generation:
                        _plan_rejected("backward", key)
                eng.plan_seen[key] = eng.ws.generation
        if self.grad_done_hook is not None:
            # data parallel: the remaining buckets go out and the compute stream is made to wait for the exchange HERE, so
            # whatever reads the gradients next (GradScaler.unscale_, gradient-norm logging, the optimizer) sees the reduced
            # values on every rank — not inside optimizer.step, which a scaler may skip on one rank only
            self.grad_done_hook()
        arena.publish_grads(wanted)

    def _logits(self, x: torch.Tensor) -> torch.Tensor:
        hip.require_device(x, "model.forward")
        if x.dim() != 4 or x.dtype != torch.float32:
            raise RuntimeError(f"expected a float32 NCHW image batch, got {tuple(x.shape)} {x.dtype}")
        if x.shape[0] == 0 or x.shape[1] != 3:
            raise RuntimeError(f"expected a non-empty batch of 3-channel images, got {tuple(x.shape)}")
        x = x.contiguous()
        # every launch below goes to the current stream of the CURRENT device (hip.stream()): make that the images' device
        with torch.cuda.device(x.device):
            self._active = self._engine(x.device, self._compute_dtype())
            self._fwd_token += 1
            params = [p for p in self.parameters() if p.requires_grad]
            if torch.is_grad_enabled() and self.training and params:
                return _NetFn.apply(self, x, *params)
            with torch.no_grad():
                return self._forward_impl(x)


class SingletaskClassifier(_HipClassifier):
    """Single task classification model (model.py:17-85)."""

    def __init__(self, cfg_model: dict, classes: list):
        super().__init__(cfg_model)
        self.classifier = nn.Sequential(
            nn.Dropout(cfg_model["classifier_dropout"]),
            nn.Linear(self.emb_size, len(classes)),
        )
        self.initialize_classifier(strategy=cfg_model["classifier_initialization"])

    def initialize_classifier(self, strategy="kaiming_normal_"):
        self._init_params(self.classifier.parameters(), strategy)

    def _heads(self):
        return [self.classifier[1]]

    def _classifier_dropout_p(self):
        return float(self.classifier[0].p)

    def forward(self, x: torch.Tensor):
        return self._logits(x)


class MultitaskClassifier(_HipClassifier):
    """Shared backbone + one Dropout->Linear head per task (model.py:88-159); heads run as one fused GEMM."""

    def __init__(self, cfg_model: dict, classes: dict):
        super().__init__(cfg_model)
        self.classifier = nn.ModuleDict()
        for target_name in classes:
            self.classifier[target_name] = nn.Sequential(
                nn.Dropout(cfg_model["classifier_dropout"]),
                nn.Linear(self.emb_size, len(classes[target_name])),
            )
        self.initialize_classifiers(strategy=cfg_model["classifier_initialization"])

    def initialize_classifiers(self, strategy="kaiming_normal_"):
        for head in self.classifier.values():
            self._init_params(head.parameters(), strategy)

    def _heads(self):
        return [head[1] for head in self.classifier.values()]

    def _classifier_dropout_p(self):
        return max(float(head[0].p) for head in self.classifier.values())

    def forward(self, x: torch.Tensor):
        fused = self._logits(x)
        out, lo = {}, 0
        for name, head in self.classifier.items():
            n = head[1].out_features
            out[name] = fused[:, lo:lo + n]
            lo += n
        return out


def get_model(cfg_model, classes, device="cpu", compile: bool = False):
    if cfg_model.get("scripted", False):
        model = torch.jit.load(cfg_model["checkpoint"], map_location="cpu")
    else:
        if cfg_model["task"] == "single":
            model = SingletaskClassifier(cfg_model, classes)
        elif cfg_model["task"] == "multi":
            model = MultitaskClassifier(cfg_model, classes)
        chkpt = cfg_model.get("checkpoint", None)
        if chkpt is not None:
            model.load_state_dict(torch.load(chkpt, map_location="cpu"))
    model.to(device)
    # `compile` is accepted for signature parity (model.py:174-175); the HIP engine has no tracing compiler.
    return model
