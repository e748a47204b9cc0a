"""ctypes binding of libnkbhip.so (include/nkbhip.h) — the only compute backend of this package.

There is deliberately no CPU or eager-torch fallback: if the shared library is missing or a call
fails, the caller gets a RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import torch

F32, BF16 = 0, 1
_TORCH_DT = {torch.float32: F32, torch.bfloat16: BF16}

_LIB = None
_LIB_PATH = Path(__file__).resolve().parents[1] / "lib" / "libnkbhip.so"

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t

_SIGS = {
    "nkb_version": (i32, []),
    "nkb_last_error": (C.c_char_p, []),
    "nkb_kernel_launches": (i64, [i32, i32]),
    "nkb_conv_gemm": (i32, [i32, i32, vp, vp, vp, vp, vp, vp] + [i32] * 18 + [vp, vp]),
    "nkb_conv_gemm_stat_tiles": (i32, [i32, i32, i32]),
    "nkb_conv_wgrad": (i32, [i32, vp, vp, vp, vp] + [i32] * 13 + [vp, i64, vp]),
    "nkb_conv_wgrad_assign": (i32, [i32, vp, vp, vp, vp] + [i32] * 13 + [vp, i64, vp]),
    "nkb_conv_wgrad_workspace_floats": (i64, [i32] * 11),
    "nkb_stem_wgrad_workspace_floats": (i64, [i32] * 5),
    "nkb_bn_finalize": (i32, [vp, i32, i32, i64, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp]),
    "nkb_bn_apply": (i32, [i32, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp]),
    "nkb_bn_backward": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp, sz, vp]),
    "nkb_bn_stats_floats": (sz, [i32, i32]),
    "nkb_bn_backward_workspace_floats": (sz, [i64, i32]),
    "nkb_conv_dgrad_bn": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32] + [i32] * 13 + [vp]),
    "nkb_convp_tiles": (i32, [i32] * 13),
    "nkb_conv1p_tiles": (i32, [i32, i64, i32, i32, i32, i32]),
    "nkb_conv1p_fwd": (i32, [i32, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "nkb_rowres_reserve_cus": (None, [i32]),
    "nkb_rowres_reserved_cus": (i32, []),
    "nkb_convp_config": (None, [i32, i32]),
    "nkb_convp_fwd": (i32, [i32, vp, vp, vp, vp] + [i32] * 8 + [vp]),
    "nkb_convp_dgrad_bn": (i32, [i32] + [vp] * 8 + [i32] * 8 + [vp]),
    "nkb_conv_dgrad_s2class": (i32, [i32] + [vp] * 9 + [i32] * 14 + [vp]),
    "nkb_bn_backward_from_stats": (i32, [i32, vp, vp, vp, i32, vp, vp, vp, i64, i32, vp, vp, vp, vp, vp]),
    "nkb_gram_bn_stats": (i32, [i32, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nkb_bn_apply_gram": (i32, [i32, vp, vp, vp, vp, i64, i32, vp, vp, sz, vp]),
    "nkb_bn_apply_gram_workspace_floats": (sz, [i64, i32]),
    "nkb_conv_affine_residual": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp] + [i32] * 13 + [vp]),
    "nkb_gram_bn_backward": (i32, [i32, vp, vp, vp, vp, vp, i32, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "nkb_gram_k1w": (i32, [i32, vp, vp, i32, i32, vp, vp]),
    "nkb_conv_dgrad_bn_add": (i32, [i32, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "nkb_gram_bn_backward_workspace_floats": (sz, [i32, i32]),
    "nkb_gram_fold2": (i32, [i32, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "nkb_conv_cat_relu_bits": (i32, [i32, vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, i64, i32, i32, vp]),
    "nkb_conv_cat_bias": (i32, [i32, vp, i32, i32, vp, i32, i32, vp, vp, vp, i64, i32, i32, vp]),
    "nkb_conv_dgrad_bn_cat": (i32, [i32, vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "nkb_wprep_multi": (i32, [i32, vp, vp, i32, i32, vp, vp]),
    "nkb_wprep_block_elems": (i32, []),
    "nkb_wprep_job_blocks": (i64, [i32, i32, i32, i32, i32]),
    "nkb_stem_pack": (i32, [i32, vp, vp, i32, i32, i32, i32, vp]),
    "nkb_stemp_tiles": (i32, [i32, i32, i32, i32, i32]),
    "nkb_stemp_wgrad_workspace_floats": (i64, [i32, i32, i32, i32, i32]),
    "nkb_stemp_wgrad": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, vp]),
    "nkb_gramr_workspace_floats": (i64, [i32, i64, i32, i32]),
    "nkb_gramr": (i32, [i32, vp, i32, vp, i32, vp, i64, i32, i32, i32, vp, i64, vp]),
    "nkb_stemp_conv": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "nkb_stem_wprep": (i32, [i32, vp, vp, i32, i32, vp]),
    "nkb_stem_weight_cols": (i32, [i32]),
    "nkb_stem_conv": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "nkb_stem_wgrad": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, vp]),
    "nkb_stem_wfold": (i32, [i32, vp, vp, i32, i32, vp]),
    "nkb_maxpool3x3s2": (i32, [i32, i32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "nkb_bn_relu_maxpool": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, vp]),
    "nkb_bn_relu_maxpool_sel": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, i32, i32, i32, i32, vp]),
    "nkb_bn_relu_maxpool_workspace_floats": (sz, [i32, i32, i32, i32]),
    "nkb_avgpool": (i32, [i32, i32, vp, vp, i32, i32, i32, vp]),
    "nkb_im2row": (i32, [i32, vp, vp] + [i32] * 9 + [vp]),
    "nkb_wprep": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "nkb_add2d": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "nkb_colsum": (i32, [i32, vp, vp, i32, i32, i32, vp]),
    "nkb_pad_cast": (i32, [i32, vp, vp, i32, i32, i32, i32, f32, vp]),
    "nkb_gemm_batched": (i32, [i32, vp, vp, vp] + [i32] * 8 + [i64] * 6 + [i32, vp]),
    "nkb_gemm_tn_batched": (i32, [i32, vp, vp, vp] + [i32] * 8 + [i64] * 6 + [vp]),
    "nkb_linear_gelu": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "nkb_linear_gelu_fused_ok": (i32, [i32, i32, i32, i32]),
    "nkb_linear_residual_scaled": (i32, [i32, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, vp]),
    "nkb_layernorm": (i32, [i32, i32, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, i64, vp, vp, i32, i32, f32, vp, vp, vp, i32, vp, i32, vp, vp]),
    "nkb_layernorm_workspace_floats": (sz, [i32]),
    "nkb_layernorm_param_reduce": (i32, [vp, i32, i32, i32, vp, vp, vp, vp]),
    "nkb_gelu": (i32, [i32, vp, vp, vp, i64, vp]),
    "nkb_splitk_reduce": (i32, [i32, vp, i32, i32, i32, vp, i32, vp, vp, vp]),
    "nkb_wfold": (i32, [i32, vp, vp, vp, i32, i32, vp]),
    "nkb_image_prep": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, f32, vp]),
    "nkb_relu6": (i32, [i32, vp, vp, vp, i64, vp]),
    "nkb_gelu_fwd_dgelu": (i32, [i32, vp, vp, vp, i64, vp]),
    "nkb_scale_rows": (i32, [i32, vp, vp, vp, vp, i32, i64, vp]),
    "nkb_attn_softmax": (i32, [i32, i32, vp, i32, vp, vp, i32, i64, i32, f32, vp]),
    "nkb_attn_forward": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp]),
    "nkb_attn_backward_ds": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, vp, i64, vp]),
    "nkb_attn_backward": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, vp]),
    "nkb_head_transpose": (i32, [i32, vp, i32, i64, i64, i32, i32, vp, i32, i32, i32, vp]),
    "nkb_vit_assemble": (i32, [i32, i32, vp, vp, vp, vp, i32, i32, i32, vp]),
    "nkb_dropout": (i32, [i32, i32, vp, vp, vp, vp, i64, f32, C.c_ulonglong, vp]),
    "nkb_colsum2d": (i32, [i32, vp, vp, i64, i32, i64, vp, vp]),
    "nkb_loss_forward": (i32, [i32, vp, i32, vp, i32, i32, vp, f32, i64, vp, i32, vp, vp, vp, i32, vp]),
    "nkb_loss_row_state_bytes": (sz, [i32]),
    "nkb_loss_backward": (i32, [vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, i32, vp]),
    "nkb_optim_step": (i32, [i32, vp, vp, vp, vp, vp, i64] + [f32] * 10 + [vp, vp]),
    "nkb_grad_unscale_check": (i32, [vp, i64, vp, vp, vp]),
    "nkb_scaler_update": (i32, [vp, vp, vp, vp, f32, f32, i32, vp]),
    "nkb_bucket_sum_bf16": (i32, [vp, i64, i32, vp, vp, i64, vp]),
    "nkb_segment_sumsq": (i32, [vp, vp, i32, vp, vp]),
    "nkb_gemm8p_config": (None, [i32, i32, i32]),
    "nkb_gemm8p_ragged": (None, [i32]),
    "nkb_fp8_quantize": (i32, [i32, i32, vp, i64, vp, vp, vp]),
    "nkb_fp8_amax": (i32, [i32, vp, i64, vp, vp]),
    "nkb_fp8_scale_update": (i32, [vp, i32, vp]),
    "nkb_fp8_job_blocks": (i64, [i64]),
    "nkb_fp8_multi": (i32, [i32, vp, i32, i64, vp]),
    "nkb_fp8_quantize_colsum_workspace_floats": (i64, [i64, i32]),
    "nkb_fp8_quantize_colsum": (i32, [i32, vp, i64, i32, i64, vp, vp, vp, vp, vp, i32, vp]),
    "nkb_wgrad_fp8_workspace_floats": (i64, [i32, i32, i32]),
    "nkb_wgrad_fp8": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i64, vp]),
    "nkb_gemm_fp8": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp] + [i32] * 8 + [vp]),
    "nkb_prof_enable": (None, [i32]),
    "nkb_prof_collect": (i32, [vp, vp, vp, vp, i32]),
    "nkb_prof_collect_raw": (i32, [vp, vp, vp, vp, i32]),
    "nkb_kernel_name": (C.c_char_p, [i32]),
    "nkb_plan_fn_count": (i32, []),
    "nkb_plan_fn_name": (C.c_char_p, [i32]),
    "nkb_plan_fn_args": (i32, [i32]),
    "nkb_plan_max_args": (i32, []),
    "nkb_plan_entry_bytes": (sz, []),
    "nkb_plan_run": (i32, [vp, i32, vp]),
}


def lib_path() -> Path:
    return Path(os.environ.get("NKBHIP_LIB", _LIB_PATH))


def load():
    """Load libnkbhip.so once; raise loudly when it is absent (no fallback exists)."""
    global _LIB
    if _REC is not None:
        return _REC_LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not path.exists():
        raise RuntimeError(
            f"libnkbhip.so not found at {path}: build it with `make -C nkb-classification_amd/csrc` "
            "(or __graft_entry__.build()). This package has no CPU / eager fallback."
        )
    lib = C.CDLL(str(path))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def exported_symbols():
    return sorted(_SIGS)


# ---- launch plans --------------------------------------------------------------------------------------------------------
# One train step is ~600 kernel launches, each reached through a few layers of Python (geometry dicts, workspace lookups,
# pointer extraction, ctypes marshalling): 17 ms of host time per ResNet-50 step against 21 ms of GPU time (BENCH_r01).  All
# of it is a pure function of (model, shapes, mode): every buffer lives in the persistent workspace / parameter arena, so the
# C-ABI calls of one step can be RECORDED once — function pointer + argument tuple + the stream / event operations between
# them — and replayed by a flat loop.  What changes from step to step is patched in: pointers into the caller's tensors
# (input images, logits, logits gradient) and dropout seeds.
_REC = None            # list of plan entries while recording
_REC_LIB = None
_PURE = frozenset({"nkb_kernel_launches", "nkb_linear_gelu_fused_ok", "nkb_version", "nkb_last_error", "nkb_conv_gemm_stat_tiles", "nkb_convp_tiles", "nkb_conv1p_tiles", "nkb_stemp_tiles", "nkb_stemp_wgrad_workspace_floats", "nkb_gramr_workspace_floats", "nkb_bn_stats_floats",
                   "nkb_bn_backward_workspace_floats", "nkb_wprep_block_elems", "nkb_wprep_job_blocks", "nkb_stem_weight_cols",
                   "nkb_bn_relu_maxpool_workspace_floats", "nkb_layernorm_workspace_floats", "nkb_loss_row_state_bytes",
                   "nkb_conv_wgrad_workspace_floats", "nkb_stem_wgrad_workspace_floats", "nkb_kernel_name",
                   "nkb_prof_enable", "nkb_prof_collect", "nkb_prof_collect_raw", "nkb_gemm8p_config", "nkb_gemm8p_ragged", "nkb_convp_config", "nkb_rowres_reserve_cus", "nkb_rowres_reserved_cus",
                   "nkb_fp8_job_blocks", "nkb_wgrad_fp8_workspace_floats",
                   "nkb_fp8_quantize_colsum_workspace_floats",
                   "nkb_gram_bn_backward_workspace_floats", "nkb_bn_apply_gram_workspace_floats",
                   "nkb_plan_fn_count", "nkb_plan_fn_name", "nkb_plan_fn_args", "nkb_plan_max_args", "nkb_plan_entry_bytes",
                   "nkb_plan_run"})


class Seed(int):
    """A dropout seed drawn from torch's host generator; a recorded plan draws a fresh one at every replay."""


def fresh_seed() -> "Seed":
    return Seed(int(torch.randint(0, 2 ** 62, (1,)).item()))


class _RecLib:
    def __getattr__(self, name):
        fn = getattr(_LIB, name)
        if name in _PURE:
            return fn

        def call(*args):
            _REC.append(["call", fn, name, list(args)])
            return fn(*args)
        return call


# The replay itself runs in C (csrc/plan.hip: nkb_plan_run walks a flat table of {function id, arguments}); Python only
# patches the few per-step values and steps from segment to segment — a segment ends where a Python-side operation sits
# between launches (the DDP bucket hooks, a counter bump).  NKB_PLAN_C=0 keeps the old per-entry ctypes loop (A/B, debugging).
_PLAN_C = os.environ.get("NKB_PLAN_C", "1") != "0"
_PLAN_MAX_ARGS = 32
OP_EVENT_RECORD, OP_STREAM_WAIT_EVENT, OP_MEMSET = -1, -2, -3


class _PlanArg(C.Union):
    _fields_ = [("p", C.c_void_p), ("i", C.c_longlong), ("f", C.c_float)]


class _PlanEntry(C.Structure):
    _fields_ = [("fn", C.c_int), ("nargs", C.c_int), ("a", _PlanArg * _PLAN_MAX_ARGS)]


_PLAN_FN = None        # {entry point name: function id of csrc/plan_dispatch.inc}


def _plan_fn_ids():
    global _PLAN_FN
    if _PLAN_FN is None:
        lib = _LIB if _LIB is not None else load()
        if lib.nkb_plan_max_args() != _PLAN_MAX_ARGS or lib.nkb_plan_entry_bytes() != C.sizeof(_PlanEntry):
            raise RuntimeError("nkbhip: NkbPlanEntry layout of libnkbhip.so differs from the binding's (rebuild the library)")
        ids = {}
        for k in range(lib.nkb_plan_fn_count()):
            name = lib.nkb_plan_fn_name(k).decode()
            if name not in _SIGS or lib.nkb_plan_fn_args(k) != len(_SIGS[name][1]):
                raise RuntimeError(f"nkbhip: plan dispatch table is stale for {name} (python scripts/gen_plan_dispatch.py; rebuild)")
            ids[name] = k
        _PLAN_FN = ids
    return _PLAN_FN


class Plan:
    __slots__ = ("segments", "dyn_names", "keep", "entries")

    def __init__(self, segments, dyn_names, keep, entries):
        # entries: number of recorded operations (diagnostics); keep: the torch events / streams / tensors the table points at
        self.segments, self.dyn_names, self.keep, self.entries = segments, dyn_names, keep, entries


_REC_MAIN = None       # the compute stream of the step being recorded


def record_begin():
    global _REC, _REC_LIB, _REC_MAIN
    load()
    _REC_MAIN = torch.cuda.current_stream()
    _REC, _REC_LIB = [], _RecLib()


def record_abort():
    global _REC
    _REC = None


def record_end(dynamic: dict) -> Plan:
    """dynamic: {name: tensor} — every recorded pointer argument that points into one of these tensors becomes a slot that
    replay() fills from the tensor passed under the same name."""
    global _REC
    rec, _REC = _REC, None
    ranges = [(name, t.data_ptr(), t.data_ptr() + max(t.numel() * t.element_size(), 1)) for name, t in dynamic.items()]
    ids = _plan_fn_ids() if _PLAN_C else {}
    segments, keep = [], []
    cur = []            # entries of the C segment being built: (fn id, [(kind, value)], name, patches)

    def flush():
        if not cur:
            return
        arr = (_PlanEntry * len(cur))()
        patches, names = [], []
        for k, (fid, vals, name, pt) in enumerate(cur):
            e = arr[k]
            e.fn, e.nargs = fid, len(vals)
            for ai, (kind, v) in enumerate(vals):
                if kind == "f":
                    e.a[ai].f = float(v)
                elif kind == "p":
                    e.a[ai].p = v
                else:
                    e.a[ai].i = int(v)
            names.append(name)
            patches += [(k, ai, dn, off) for ai, dn, off in pt]
        segments.append((0, arr, len(cur), tuple(patches), tuple(names)))
        cur.clear()

    for e in rec:
        if e[0] == "py":
            flush()
            segments.append((2, e[1], None, None, None))
            continue
        if e[0] == "op":
            _, code, vals, alive = e
            keep.append(alive)
            if _PLAN_C:
                kinds = ("p", "i", "p") if code == OP_MEMSET else ("p", "p")
                cur.append((code, list(zip(kinds, vals)), "stream operation %d" % code, ()))
            else:
                segments.append((2, _op_closure(code, alive), None, None, None))
            continue
        _, fn, name, args = e
        patches = []
        for i, a in enumerate(args):
            if isinstance(a, Seed):
                patches.append((i, None, 0))
            elif isinstance(a, int) and a >= (1 << 32):
                for dn, lo, hi in ranges:
                    if lo <= a < hi:
                        patches.append((i, dn, a - lo))
                        break
        if name in ids and len(args) <= _PLAN_MAX_ARGS:
            types = _SIGS[name][1]
            vals = [("f" if t is f32 else ("p" if t is vp else "i"), (0 if a is None and t is not vp else a)) for t, a in zip(types, args)]
            cur.append((ids[name], vals, name, patches))
        else:
            flush()
            segments.append((1, fn, list(args), tuple(patches), name))
    flush()
    return Plan(segments, tuple(dynamic), keep, len(rec))


def _op_closure(code, alive):
    """Python form of a recorded stream operation (NKB_PLAN_C=0)."""
    if code == OP_EVENT_RECORD:
        ev, st = alive
        return lambda: ev.record(st)
    if code == OP_STREAM_WAIT_EVENT:
        ev, st = alive
        return lambda: st.wait_event(ev)
    t, st = alive

    def memset():
        with torch.cuda.stream(st):
            t.zero_()
    return memset


_FAILED = C.c_int(0)


def replay(plan: Plan, dynamic: dict):
    base = {name: t.data_ptr() for name, t in dynamic.items()}
    for kind, a, b, patches, names in plan.segments:
        if kind == 0:
            for ei, ai, dn, off in patches:
                a[ei].a[ai].i = base[dn] + off if dn is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
            if _LIB.nkb_plan_run(a, b, C.byref(_FAILED)):
                raise RuntimeError(f"nkbhip {names[_FAILED.value]} failed during plan replay: {_LIB.nkb_last_error().decode()}")
        elif kind == 1:
            for i, dn, off in patches:
                b[i] = base[dn] + off if dn is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
            if a(*b):
                raise RuntimeError(f"nkbhip {names} failed during plan replay: {_LIB.nkb_last_error().decode()}")
        else:
            a()


# ---- stream operations between launches, recordable (a host_op closure would end the C segment) --------------------------------
def event_record(ev: "torch.cuda.Event", st: "torch.cuda.Stream"):
    ev.record(st)
    if _REC is not None:
        _REC.append(["op", OP_EVENT_RECORD, (ev.cuda_event, st.cuda_stream), (ev, st)])


def stream_wait_event(st: "torch.cuda.Stream", ev: "torch.cuda.Event"):
    st.wait_event(ev)
    if _REC is not None and ev.cuda_event:          # (an event that was never recorded: torch's wait is a no-op, so is the plan's)
        _REC.append(["op", OP_STREAM_WAIT_EVENT, (st.cuda_stream, ev.cuda_event), (ev, st)])


def stream_wait_stream(st: "torch.cuda.Stream", other: "torch.cuda.Stream"):
    """st waits for everything enqueued on `other` so far (torch's Stream.wait_stream with a recordable event)."""
    ev = torch.cuda.Event()
    event_record(ev, other)
    stream_wait_event(st, ev)


def zero_(t: torch.Tensor):
    """t.zero_() on the current stream, as a recordable memset."""
    t.zero_()
    if _REC is not None:
        if not t.is_contiguous():
            raise RuntimeError("hip.zero_: recorded memsets need a contiguous tensor")
        st = torch.cuda.current_stream()
        _REC.append(["op", OP_MEMSET, (t.data_ptr(), t.numel() * t.element_size(), st.cuda_stream), (t, st)])


def recording() -> bool:
    return _REC is not None


def host_op(fn):
    """A torch-side operation between launches (memset, counter bump, stream / event call): run now, and re-run at this
    point of the sequence when the plan is replayed — on the stream that is current now."""
    global _REC
    if _REC is not None:
        cur = torch.cuda.current_stream()
        if cur == _REC_MAIN:
            _REC.append(["py", fn])
        else:
            def on_stream(fn=fn, cur=cur):
                with torch.cuda.stream(cur):
                    fn()
            _REC.append(["py", on_stream])
        # fn itself is what the plan re-runs: library calls it makes now (the bf16 gradient exchange of parallel.GradReducer
        # casts and sums through libnkbhip on temporaries) must not ALSO be recorded as table entries — replayed, they would
        # run a second time on pointers that were freed when fn returned
        rec, _REC = _REC, None
        before = _device_allocs()
        try:
            fn()
        finally:
            _REC = rec
            # what fn allocated is fn's own business (it runs again, allocating afresh, at every replay): the model's "the recorded
            # run allocated nothing" test (model.py) must not count it, or a plan with a bucket hook in it would never be kept
            global host_op_allocs
            host_op_allocs += _device_allocs() - before
        return
    fn()


host_op_allocs = 0     # device allocations made inside host_op closures while a plan was being recorded (monotone)


def _device_allocs() -> int:
    if not torch.cuda.is_available():
        return 0
    return int(torch.cuda.memory_stats(torch.cuda.current_device()).get("allocation.all.allocated", 0))


def kernel_launches(which: str, reset: bool = False) -> int:
    """Launch count of a specialised kernel family: gemm8p, wgrad8p, wgrad3x3, wgrad8f, gram_conv, gram_bn_apply, convp, conv1p, stemp, gramr, wgradr."""
    idx = {"gemm8p": 0, "wgrad8p": 1, "wgrad3x3": 2, "wgrad8f": 3, "gram_conv": 4, "gram_bn_apply": 5, "convp": 6, "conv1p": 7,
           "stemp": 8, "gramr": 9, "wgradr": 10}[which]
    return int(load().nkb_kernel_launches(idx, int(reset)))


def dt(t) -> int:
    d = t if isinstance(t, torch.dtype) else t.dtype
    if d not in _TORCH_DT:
        raise RuntimeError(f"nkbhip: unsupported dtype {d}")
    return _TORCH_DT[d]


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    # the current stream of the CURRENT device: entry points run under torch.cuda.device(<tensor device>) (model._logits,
    # FusedOptimizer.step, train.py's set_device), so this is the stream of the device that holds the operands
    return torch.cuda.current_stream().cuda_stream


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"nkbhip {what} failed (code {rc}): {load().nkb_last_error().decode()}")


def require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the HIP engine runs on cuda devices only "
                           "(there is no CPU fallback)")


# ------------------------------------------------------------------ thin typed wrappers ----
def conv_gemm(dtype, mode, x, w, y, *, N, H, W, Cin, ldx, P, Q, Cout, ldy, R=1, S=1, stride=1, pad=0, add=None,
              ldadd=0, bias=None, stats=None, relu=False, out_f32=False, add_hw=(0, 0), add_bits=None):
    check(load().nkb_conv_gemm(dtype, mode, ptr(x), ptr(w), ptr(y), ptr(add), ptr(bias), ptr(stats), N, H, W, Cin, ldx,
                               P, Q, Cout, ldy, ldadd, R, S, stride, pad, int(relu), int(out_f32), add_hw[0], add_hw[1],
                               ptr(add_bits), stream()), "conv_gemm")


def stat_tiles(dtype, M, Cout):
    return load().nkb_conv_gemm_stat_tiles(dtype, M, Cout)


def convp_tiles(dtype, kind, *, N, H, W, Cin, ldx, Cout, ldy, R, S, stride, pad) -> int:
    """Partial-sum rows of the row-balanced 3x3 core for this launch (kind 0 forward, 1 data gradient); 0 = not eligible."""
    return int(load().nkb_convp_tiles(dtype, kind, N, H, W, Cin, ldx, Cout, ldy, R, S, stride, pad))


def convp_config(on: bool = True, tc128: bool = False, c64: bool = True, c64_dgrad: bool = False, conv1p: bool = True, stemp: bool = True,
                 gramr: bool = True):
    """Envelope of the row-resident kernel family (convp / conv1p / stemp / gramr); the defaults are the train step's."""
    load().nkb_convp_config(int(on), int(tc128) | (2 if c64 else 0) | (4 if c64_dgrad else 0) | (0 if conv1p else 16) | (0 if stemp else 32)
                            | (0 if gramr else 64))


_geometry_epoch = 0


def geometry_epoch() -> int:
    """Bumped whenever a setting that sizes grids (and through them stats / slab buffers) changes: runtime.Workspace.generation
    carries it, so launch plans recorded under the old setting are dropped instead of replayed."""
    return _geometry_epoch


def rowres_reserve_cus(cus: int):
    """Data-parallel runs: CUs the row-resident family's backward kernels (convp data gradient, gramr, wgradr) leave to the
    collective.  Changing it after a step has run is allowed: recorded plans are invalidated and the next eager pass re-sizes the
    partial-sum / slab buffers; a stale buffer handed to the library is refused by the launch itself."""
    global _geometry_epoch
    lib = load()
    if int(lib.nkb_rowres_reserved_cus()) != max(0, min(128, int(cus))):
        _geometry_epoch += 1
    lib.nkb_rowres_reserve_cus(int(cus))


def conv1p_tiles(dtype, M, Cin, ldx, Cout, ldy) -> int:
    """Partial-sum rows of conv1p_fwd for this 1x1 shape; 0: not eligible (use conv_gemm)."""
    return int(load().nkb_conv1p_tiles(dtype, M, Cin, ldx, Cout, ldy))


def conv1p_fwd(dtype, x, w, y, stats, *, M, Cin, ldx, Cout, ldy):
    check(load().nkb_conv1p_fwd(dtype, ptr(x), ptr(w), ptr(y), ptr(stats), M, Cin, ldx, Cout, ldy, stream()), "conv1p_fwd")


def convp_fwd(dtype, x, w, y, stats, *, N, H, W, Cin, ldx, Cout, ldy, tiles):
    """tiles: the partial-sum rows `stats` was sized for (convp_tiles at that time); the launch is refused if its grid differs."""
    check(load().nkb_convp_fwd(dtype, ptr(x), ptr(w), ptr(y), ptr(stats), N, H, W, Cin, ldx, Cout, ldy, int(tiles), stream()), "convp_fwd")


def convp_dgrad_bn(dtype, dy, w, g_masked, c, scale, shift, mean, stats, *, N, H, W, Cin, ldx, Cout, ldy, tiles):
    check(load().nkb_convp_dgrad_bn(dtype, ptr(dy), ptr(w), ptr(g_masked), ptr(c), ptr(scale), ptr(shift), ptr(mean), ptr(stats),
                                    N, H, W, Cin, ldx, Cout, ldy, int(tiles), stream()), "convp_dgrad_bn")


def conv_wgrad(dtype, dy, x, dw, *, N, H, W, Cin, ldx, P, Q, Cout, lddy, R=1, S=1, stride=1, pad=0, dbias=None,
               workspace=None, assign=False):
    """workspace: fp32 scratch of at least conv_wgrad_workspace(...) floats -> deterministic two-stage accumulation; None ->
    fp32 atomics.  assign: dw / dbias are overwritten instead of accumulated into (needs the workspace)."""
    fn = load().nkb_conv_wgrad_assign if assign else load().nkb_conv_wgrad
    check(fn(dtype, ptr(dy), ptr(x), ptr(dw), ptr(dbias), N, H, W, Cin, ldx, P, Q, Cout, lddy, R, S, stride, pad,
                                ptr(workspace), workspace.numel() if workspace is not None else 0, stream()), "conv_wgrad")


def conv_wgrad_workspace(dtype, *, N, P, Q, Cin, Cout, R=1, S=1, stride=1, pad=0, has_bias=False) -> int:
    return int(load().nkb_conv_wgrad_workspace_floats(dtype, N, P, Q, Cin, Cout, R, S, stride, pad, int(has_bias)))


def stem_wgrad_workspace(dtype, N, H, W, Cout) -> int:
    return int(load().nkb_stem_wgrad_workspace_floats(dtype, N, H, W, Cout))


def bn_finalize(partials, tiles, C_, count, gamma, beta, rm, rv, momentum, eps, training, scale, shift, mean, invstd):
    check(load().nkb_bn_finalize(ptr(partials), tiles, C_, count, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), momentum, eps,
                                 int(training), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), stream()), "bn_finalize")


def bn_apply(dtype, x, res, y, scale, shift, rows, C_, relu, relu_bits=None, res_scale=None, res_shift=None):
    check(load().nkb_bn_apply(dtype, ptr(x), ptr(res), ptr(y), ptr(scale), ptr(shift), rows, C_, int(relu),
                              ptr(relu_bits), ptr(res_scale), ptr(res_shift), stream()), "bn_apply")


def bn_stats_floats(tiles, C_):
    return load().nkb_bn_stats_floats(tiles, C_)


def bn_backward(dtype, dy, x, yact, mean, invstd, gamma, rows, C_, dgamma, dbeta, dx, dy_masked, workspace,
                fscale=None, fshift=None, relu_bits=None):
    check(load().nkb_bn_backward(dtype, ptr(dy), ptr(x), ptr(yact), ptr(relu_bits), ptr(fscale), ptr(fshift), ptr(mean),
                                 ptr(invstd),
                                 ptr(gamma), rows, C_,
                                 ptr(dgamma), ptr(dbeta), ptr(dx), ptr(dy_masked), ptr(workspace), workspace.numel(),
                                 stream()), "bn_backward")


def conv_dgrad_bn(dtype, dy, w, g_masked, c, scale, shift, mean, stats, *, N, H, W, Cin, ldx, P, Q, Cout, ldy, R, S,
                  stride, pad, relu_bits=None, add=None, ldadd=0, add_bits=None, add_hw=(0, 0)):
    check(load().nkb_conv_dgrad_bn(dtype, ptr(dy), ptr(w), ptr(g_masked), ptr(c), ptr(scale), ptr(shift), ptr(mean),
                                   ptr(stats), ptr(relu_bits), ptr(add), ldadd, ptr(add_bits), add_hw[0], add_hw[1],
                                   N, H, W, Cin, ldx, P, Q, Cout, ldy, R, S, stride, pad, stream()),
          "conv_dgrad_bn")


def conv_dgrad_s2class(dtype, dy, w_class, y, add, c, scale, shift, mean, stats, N, Hdy, Wdy, K, ldx, Hout, Wout, C_, ldy,
                       ldadd, ph, pw, add_h=0, add_w=0):
    check(load().nkb_conv_dgrad_s2class(dtype, ptr(dy), ptr(w_class), ptr(y), ptr(add), ptr(c), ptr(scale), ptr(shift),
                                        ptr(mean), ptr(stats), N, Hdy, Wdy, K, ldx, Hout, Wout, C_, ldy, ldadd, ph, pw,
                                        add_h, add_w, stream()), "conv_dgrad_s2class")


def bn_backward_from_stats(dtype, g, x, stats, tiles, mean, invstd, gamma, rows, C_, dgamma, dbeta, dx, sums):
    check(load().nkb_bn_backward_from_stats(dtype, ptr(g), ptr(x), ptr(stats), tiles, ptr(mean), ptr(invstd), ptr(gamma),
                                            rows, C_, ptr(dgamma), ptr(dbeta), ptr(dx), ptr(sums), stream()),
          "bn_backward_from_stats")


def gram_bn_stats(dtype, w, gram, colsum, count, Cin, Cout, gamma, beta, rm, rv, momentum, eps, cov, mu, T, scale, shift, mean, invstd):
    check(load().nkb_gram_bn_stats(dtype, ptr(w), ptr(gram), ptr(colsum), count, Cin, Cout, ptr(gamma), ptr(beta), ptr(rm), ptr(rv),
                                   momentum, eps, ptr(cov), ptr(mu), ptr(T), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), stream()),
          "gram_bn_stats")


def bn_apply_gram_ws(rows, C_) -> int:
    return int(load().nkb_bn_apply_gram_workspace_floats(rows, C_))


def bn_apply_gram(dtype, c, y, scale, shift, rows, C_, gram, work):
    check(load().nkb_bn_apply_gram(dtype, ptr(c), ptr(y), ptr(scale), ptr(shift), rows, C_, ptr(gram), ptr(work), work.numel(), stream()),
          "bn_apply_gram")


def conv_affine_residual(dtype, x, w, y, scale, shift, res, ldres, res_scale, res_shift, relu_bits, *, N, H, W, Cin, ldx, P, Q, Cout,
                         ldy, R=1, S=1, stride=1, pad=0):
    check(load().nkb_conv_affine_residual(dtype, ptr(x), ptr(w), ptr(y), ptr(scale), ptr(shift), ptr(res), ldres, ptr(res_scale),
                                          ptr(res_shift), ptr(relu_bits), N, H, W, Cin, ldx, P, Q, Cout, ldy, R, S, stride, pad,
                                          stream()), "conv_affine_residual")


def gram_bn_backward_ws(Cin, Cout) -> int:
    return int(load().nkb_gram_bn_backward_workspace_floats(Cin, Cout))


def gram_bn_backward(dtype, w, R, T, mu, gstats, tiles, count, Cin, Cout, gamma, mean, invstd, dgamma, dbeta, dw, wcat, cbias, coef, q=None):
    check(load().nkb_gram_bn_backward(dtype, ptr(w), ptr(R), ptr(T), ptr(mu), ptr(gstats), tiles, count, Cin, Cout, ptr(gamma), ptr(mean),
                                      ptr(invstd), ptr(dgamma), ptr(dbeta), ptr(dw), ptr(wcat), ptr(q), ptr(cbias), ptr(coef), stream()),
          "gram_bn_backward")


def gram_k1w(dtype, w, k1, Cin, Cout, out):
    check(load().nkb_gram_k1w(dtype, ptr(w), ptr(k1), Cin, Cout, ptr(out), stream()), "gram_k1w")


def conv_dgrad_bn_add(dtype, a, lda, K, q, cbias, t, ldt, g_masked, c_prev, scale, shift, mean, stats, M, Cout, ldy):
    check(load().nkb_conv_dgrad_bn_add(dtype, ptr(a), lda, K, ptr(q), ptr(cbias), ptr(t), ldt, ptr(g_masked), ptr(c_prev), ptr(scale),
                                       ptr(shift), ptr(mean), ptr(stats), M, Cout, ldy, stream()), "conv_dgrad_bn_add")


def conv_dgrad_bn_cat(dtype, g, ldg, K1, a, lda, K2, wcat, cbias, g_masked, c_prev, scale, shift, mean, stats, M, Cout, ldy):
    check(load().nkb_conv_dgrad_bn_cat(dtype, ptr(g), ldg, K1, ptr(a), lda, K2, ptr(wcat), ptr(cbias), ptr(g_masked), ptr(c_prev),
                                       ptr(scale), ptr(shift), ptr(mean), ptr(stats), M, Cout, ldy, stream()), "conv_dgrad_bn_cat")


def gram_fold2(dtype, w1, s1, K1, w2, s2, K2, Cout, out, shift1, shift2, shift_out):
    check(load().nkb_gram_fold2(dtype, ptr(w1), ptr(s1), K1, ptr(w2), ptr(s2), K2, Cout, ptr(out), ptr(shift1), ptr(shift2), ptr(shift_out),
                                stream()), "gram_fold2")


def conv_cat_relu_bits(dtype, a, lda, K1, x, ldx, K2, wf, shift, y, relu_bits, M, Cout, ldy):
    check(load().nkb_conv_cat_relu_bits(dtype, ptr(a), lda, K1, ptr(x), ldx, K2, ptr(wf), ptr(shift), ptr(y), ptr(relu_bits), M, Cout, ldy,
                                        stream()), "conv_cat_relu_bits")


def conv_cat_bias(dtype, a, lda, K1, x, ldx, K2, w, bias, y, M, Cout, ldy):
    check(load().nkb_conv_cat_bias(dtype, ptr(a), lda, K1, ptr(x), ldx, K2, ptr(w), ptr(bias), ptr(y), M, Cout, ldy, stream()), "conv_cat_bias")


def bn_backward_ws(rows, C_):
    return load().nkb_bn_backward_workspace_floats(rows, C_)


def bn_relu_maxpool(dtype, backward, c, scale, shift, mean, invstd, gamma, y_or_g, idx, dc, dgamma, dbeta, work, N, H, W, C_, xsel=None):
    """xsel [N][P][Q][C]: written by the forward call (raw conv output behind every pooled winner), read by the backward call."""
    check(load().nkb_bn_relu_maxpool_sel(dtype, int(backward), ptr(c), ptr(scale), ptr(shift), ptr(mean), ptr(invstd),
                                         ptr(gamma), ptr(y_or_g), ptr(idx), ptr(dc), ptr(dgamma), ptr(dbeta), ptr(work),
                                         work.numel() if work is not None else 0, ptr(xsel), N, H, W, C_, stream()), "bn_relu_maxpool")


def bn_relu_maxpool_ws(N, H, W, C_) -> int:
    return int(load().nkb_bn_relu_maxpool_workspace_floats(N, H, W, C_))


def maxpool(dtype, backward, src, dst, idx, N, H, W, C_):
    check(load().nkb_maxpool3x3s2(dtype, int(backward), ptr(src), ptr(dst), ptr(idx), N, H, W, C_, stream()), "maxpool")


def avgpool(dtype, backward, src, dst, N, HW, C_):
    check(load().nkb_avgpool(dtype, int(backward), ptr(src), ptr(dst), N, HW, C_, stream()), "avgpool")


def stem_pack(dtype, x, out, N, C_, H, W):
    check(load().nkb_stem_pack(dtype, ptr(x), ptr(out), N, C_, H, W, stream()), "stem_pack")


def stem_wprep(dtype, w, wp, Cout, C_):
    check(load().nkb_stem_wprep(dtype, ptr(w), ptr(wp), Cout, C_, stream()), "stem_wprep")


def stem_weight_cols(dtype) -> int:
    return int(load().nkb_stem_weight_cols(dtype))


def stem_conv(dtype, xp, wp, y, stats, N, H, W, Cout, ldy):
    check(load().nkb_stem_conv(dtype, ptr(xp), ptr(wp), ptr(y), ptr(stats), N, H, W, Cout, ldy, stream()), "stem_conv")


def gramr_workspace(dtype, M, co, ci) -> int:
    """Slab floats of gramr for R[co][ci] = g^T a over M pixels; 0: not eligible (use conv_wgrad(assign=True))."""
    return int(load().nkb_gramr_workspace_floats(dtype, M, co, ci))


def gramr(dtype, g, ldg, a, lda, R, M, co, ci, workspace, assign=True, transposed=False):
    """R = g^T a (assign: overwritten, else accumulated into; transposed: stored [ci][co] = a 1x1 weight gradient with g = the input)."""
    check(load().nkb_gramr(dtype, ptr(g), ldg, ptr(a), lda, ptr(R), M, co, ci, int(assign) | (2 if transposed else 0), ptr(workspace),
                           workspace.numel(), stream()), "gramr")


def stemp_tiles(dtype, N, H, W, Cout) -> int:
    """Partial-sum rows of stemp_conv for this stem; 0: not eligible (use stem_conv)."""
    return int(load().nkb_stemp_tiles(dtype, N, H, W, Cout))


def stemp_conv(dtype, xp, wp, y, stats, N, H, W, Cout, ldy):
    check(load().nkb_stemp_conv(dtype, ptr(xp), ptr(wp), ptr(y), ptr(stats), N, H, W, Cout, ldy, stream()), "stemp_conv")


def stemp_wgrad_workspace(dtype, N, H, W, Cout) -> int:
    """Slab floats of stemp_wgrad; 0: not eligible (use stem_wgrad)."""
    return int(load().nkb_stemp_wgrad_workspace_floats(dtype, N, H, W, Cout))


def stemp_wgrad(dtype, dy, xp, dwp, N, H, W, Cout, lddy, workspace):
    check(load().nkb_stemp_wgrad(dtype, ptr(dy), ptr(xp), ptr(dwp), N, H, W, Cout, lddy, ptr(workspace), workspace.numel(), stream()),
          "stemp_wgrad")


def stem_wgrad(dtype, dy, xp, dwp, N, H, W, Cout, lddy, workspace=None):
    check(load().nkb_stem_wgrad(dtype, ptr(dy), ptr(xp), ptr(dwp), N, H, W, Cout, lddy, ptr(workspace),
                                workspace.numel() if workspace is not None else 0, stream()), "stem_wgrad")


def stem_wfold(dtype, dwp, dw, Cout, C_):
    check(load().nkb_stem_wfold(dtype, ptr(dwp), ptr(dw), Cout, C_, stream()), "stem_wfold")


def im2row(dtype, x, col, N, Cin, H, W, R, S, stride, pad, Kp):
    check(load().nkb_im2row(dtype, ptr(x), ptr(col), N, Cin, H, W, R, S, stride, pad, Kp, stream()), "im2row")


def wprep_multi(dtype, base, jobs, njobs, total_blocks, shadow=None):
    """shadow: optional bf16 mirror of `base` (same element offsets) to read the transposing jobs' sources from."""
    check(load().nkb_wprep_multi(dtype, ptr(base), ptr(jobs), njobs, total_blocks, ptr(shadow), stream()), "wprep_multi")


def wprep_block_elems() -> int:
    return int(load().nkb_wprep_block_elems())


def wprep_job_blocks(A, B, C_, ld, mode) -> int:
    return int(load().nkb_wprep_job_blocks(A, B, C_, ld, mode))


def wprep(dtype, src, dst, A, B, C_, ld, mode):
    check(load().nkb_wprep(dtype, ptr(src), ptr(dst), A, B, C_, ld, mode, stream()), "wprep")


def add2d(src, dst, rows, cols, ld_src, ld_dst):
    check(load().nkb_add2d(ptr(src), ptr(dst), rows, cols, ld_src, ld_dst, stream()), "add2d")


def colsum(dtype, x, out, rows, C_, ld):
    check(load().nkb_colsum(dtype, ptr(x), ptr(out), rows, C_, ld, stream()), "colsum")


def pad_cast(dtype, src, dst, rows, C_, ld_src, ld_dst, mul=1.0):
    check(load().nkb_pad_cast(dtype, ptr(src), ptr(dst), rows, C_, ld_src, ld_dst, mul, stream()), "pad_cast")


def loss_forward(kind, logits, ld, target, B, C_, class_weight, gamma, ignore_index, probs, ldp, argmax, row_state, out2,
                 reduction=0):
    check(load().nkb_loss_forward(kind, ptr(logits), ld, ptr(target), B, C_, ptr(class_weight), gamma, ignore_index,
                                  ptr(probs), ldp, ptr(argmax), ptr(row_state), ptr(out2), reduction, stream()),
          "loss_forward")


def loss_backward(probs, ldp, target, row_state, out2, grad_out, B, C_, dlogits, ldd, per_row=False):
    check(load().nkb_loss_backward(ptr(probs), ldp, ptr(target), ptr(row_state), ptr(out2), ptr(grad_out), int(per_row), B,
                                   C_, ptr(dlogits), ldd, stream()), "loss_backward")


def optim_step(kind, p, g, m, v, shadow, n, lr, wd, beta1, beta2, eps, grad_scale, c0=0.0, c1=0.0, c2=0.0, c3=0.0,
               skip_flag=None):
    check(load().nkb_optim_step(kind, ptr(p), ptr(g), ptr(m), ptr(v), ptr(shadow), n, lr, wd, beta1, beta2, eps,
                                grad_scale, c0, c1, c2, c3, ptr(skip_flag), stream()), "optim_step")


def bucket_sum_bf16(parts, stride, nparts, out, out_bf16, n):
    check(load().nkb_bucket_sum_bf16(ptr(parts), stride, nparts, ptr(out), ptr(out_bf16), n, stream()), "bucket_sum_bf16")


def grad_unscale_check(g, n, scale, found_inf):
    check(load().nkb_grad_unscale_check(ptr(g), n, ptr(scale), ptr(found_inf), stream()), "grad_unscale_check")


def scaler_update(scale, tracker, found_inf, last_found_inf, growth, backoff, interval):
    check(load().nkb_scaler_update(ptr(scale), ptr(tracker), ptr(found_inf), ptr(last_found_inf), growth, backoff,
                                   interval, stream()), "scaler_update")


def segment_sumsq(x, offsets, nseg, out):
    check(load().nkb_segment_sumsq(ptr(x), ptr(offsets), nseg, ptr(out), stream()), "segment_sumsq")


def gemm8p_config(on: bool, min_tiles: int = 0, min_k: int = 0):
    load().nkb_gemm8p_config(int(on), min_tiles, min_k)


def gemm8p_ragged(on: bool):
    """The M % 256 ragged rows of a persistent GEMM launch on their own small kernel where that saves a round (tests / A-B timing)."""
    load().nkb_gemm8p_ragged(int(on))


# ---- fp8 (per-tensor scaled e4m3 / e5m2 operands, configs[4]) -----------------------------------------------------------
E4M3, E5M2 = 0, 1


def fp8_quantize(dtype, kind, src, n, state, dst):
    check(load().nkb_fp8_quantize(dtype, kind, ptr(src), n, ptr(state), ptr(dst), stream()), "fp8_quantize")


def fp8_amax(dtype, src, n, state):
    check(load().nkb_fp8_amax(dtype, ptr(src), n, ptr(state), stream()), "fp8_amax")


def fp8_scale_update(state, kind):
    check(load().nkb_fp8_scale_update(ptr(state), kind, stream()), "fp8_scale_update")


def fp8_job_blocks(n) -> int:
    return int(load().nkb_fp8_job_blocks(n))


def fp8_multi(pass_, jobs, njobs, total_blocks):
    """pass 0: amax only, 1: quantise (+ amax), 2: scales from amax, 3: clear amax — over a device job table."""
    check(load().nkb_fp8_multi(pass_, ptr(jobs), njobs, total_blocks, stream()), "fp8_multi")


def gemm_fp8(mode, xq, wq, y, M, K, N, *, deq_x, deq_w, bias=None, add=None, aux=None, aux_mode=0, yq=None, q_state=None, q_kind=0,
             row_scale=None, rows_per_sample=0, mask_out=None, mask_in=None, colsum=None, colsum_work=None, ldx=None, ldw=None, ldy=None,
             ldadd=0, relu=0):
    """aux ([M][ldy], bf16): aux_mode 0 multiplies the result by it, 1 keeps the result where 0 < aux < 6 (ReLU6 backward).
    yq / q_state / q_kind: optional fp8 copy of the result (scale q_state[0], amax into q_state[2]) for the next fp8 GEMM.
    mask_out / mask_in: the ReLU6 mask as bits ([M][N / 8] bytes, with yq); y may be None when mask_out is given.
    colsum / colsum_work (with mask_in, yq): colsum[N] += column sums of the result, workspace [M / 256][N] fp32; y may be None."""
    check(load().nkb_gemm_fp8(mode, ptr(xq), ptr(wq), ptr(y), ptr(bias), ptr(add), ptr(aux), int(aux_mode), ptr(yq), ptr(q_state),
                              int(q_kind), ptr(row_scale), int(rows_per_sample), ptr(mask_out), ptr(mask_in), ptr(colsum), ptr(colsum_work),
                              ptr(deq_x), ptr(deq_w),
                              M, K, N,
                              K if ldx is None else ldx, K if ldw is None else ldw, N if ldy is None else ldy, ldadd,
                              int(relu), stream()), "gemm_fp8")


def fp8_quantize_colsum_workspace(rows, C) -> int:
    return int(load().nkb_fp8_quantize_colsum_workspace_floats(rows, C))


def fp8_quantize_colsum(kind, src, rows, C, ld, state, dst, colsum, workspace, row_scale=None, rows_per_sample=0):
    """fp8_quantize of a [rows][C] bf16 matrix + colsum[C] += its column sums (a Linear's bias gradient when src = dY); with
    row_scale the matrix is row_scale[row // rows_per_sample] * src."""
    check(load().nkb_fp8_quantize_colsum(kind, ptr(src), rows, C, ld, ptr(state), ptr(dst), ptr(colsum), ptr(workspace),
                                         ptr(row_scale), int(rows_per_sample), stream()), "fp8_quantize_colsum")


def wgrad_fp8_workspace(M, Cin, Cout) -> int:
    """Floats of workspace nkb_wgrad_fp8 needs, -1 when the shape is outside its envelope."""
    return int(load().nkb_wgrad_fp8_workspace_floats(M, Cin, Cout))


def wgrad_fp8(gq, xq, dw, M, Cin, Cout, *, deq_g, deq_x, workspace, ldx=None, ldg=None):
    """dw += deq_g * deq_x * gq^T xq (gq e5m2 [M][Cout], xq e4m3 [M][Cin], fp32 accumulation, deterministic)."""
    check(load().nkb_wgrad_fp8(ptr(gq), ptr(xq), ptr(dw), ptr(deq_g), ptr(deq_x), M, Cin, Cin if ldx is None else ldx, Cout,
                               Cout if ldg is None else ldg, ptr(workspace), workspace.numel(), stream()), "wgrad_fp8")


def prof_enable(on: bool):
    load().nkb_prof_enable(int(on))


def prof_collect():
    n = 32
    ms = (C.c_double * n)()
    cnt = (C.c_longlong * n)()
    work = (C.c_double * n)()
    byts = (C.c_double * n)()
    k = load().nkb_prof_collect(ms, cnt, work, byts, n)
    out = {}
    for i in range(k):
        if cnt[i]:
            out[load().nkb_kernel_name(i).decode()] = dict(ms=ms[i], launches=cnt[i], work=work[i], bytes=byts[i])
    return out


def prof_collect_raw(cap: int = 1 << 16, with_bytes: bool = False):
    """[(kernel name, ms, work)] per launch, in launch order (with_bytes: + the launch's algorithmic bytes)."""
    kid = (C.c_int * cap)()
    ms = (C.c_double * cap)()
    work = (C.c_double * cap)()
    byt = (C.c_double * cap)()
    n = load().nkb_prof_collect_raw(kid, ms, work, byt, cap)
    if with_bytes:
        return [(load().nkb_kernel_name(kid[i]).decode(), ms[i], work[i], byt[i]) for i in range(n)]
    return [(load().nkb_kernel_name(kid[i]).decode(), ms[i], work[i]) for i in range(n)]


# ---- transformer ----------------------------------------------------------------------------------
def gemm_batched(dtype, x, w, y, M, N, K, ldx, ldw, ldy, outer, inner, sx, sw, sy, out_f32=False):
    check(load().nkb_gemm_batched(dtype, ptr(x), ptr(w), ptr(y), M, N, K, ldx, ldw, ldy, outer, inner, sx[0], sx[1],
                                  sw[0], sw[1], sy[0], sy[1], int(out_f32), stream()), "gemm_batched")


def gemm_tn_batched(dtype, a, b, out, M, Na, Nb, lda, ldb, ldo, outer, inner, sa, sb, so):
    check(load().nkb_gemm_tn_batched(dtype, ptr(a), ptr(b), ptr(out), M, Na, Nb, lda, ldb, ldo, outer, inner, sa[0], sa[1],
                                     sb[0], sb[1], so[0], so[1], stream()), "gemm_tn_batched")


def layernorm_fwd(dtype, x, x_stride, gamma, beta, y, y_stride, mean, rstd, rows, D, eps, yq=None, q_state=None, q_kind=0):
    """yq / q_state / q_kind: optional fp8 copy of y for the fp8 GEMM that consumes it (see fp8_quantize)."""
    check(load().nkb_layernorm(dtype, 0, ptr(x), x_stride, None, 0, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), None,
                               ptr(y), y_stride, None, None, rows, D, eps, None, ptr(yq), ptr(q_state), int(q_kind), None, 0, None,
                               stream()), "layernorm_fwd")


def layernorm_bwd(dtype, dy, dy_stride, x, x_stride, gamma, mean, rstd, add, dx, dx_stride, dgamma, dbeta, rows, D,
                  workspace=None, yq=None, q_state=None, q_kind=0, row_scale=None, rows_per_sample=0, colsum=None):
    """yq / q_state / q_kind / row_scale / colsum: fp8 copy of (row_scale *) dx and its column sums (+= colsum) for the Linear
    backward that consumes this gradient — what fp8_quantize_colsum would make of dx."""
    check(load().nkb_layernorm(dtype, 1, ptr(dy), dy_stride, ptr(x), x_stride, ptr(gamma), None, ptr(mean), ptr(rstd),
                               ptr(add), ptr(dx), dx_stride, ptr(dgamma), ptr(dbeta), rows, D, 0.0, ptr(workspace),
                               ptr(yq), ptr(q_state), int(q_kind), ptr(row_scale), int(rows_per_sample), ptr(colsum),
                               stream()), "layernorm_bwd")


def layernorm_param_reduce(workspace, rows, D, planes, dgamma, dbeta, colsum=None):
    """Second half of layernorm_bwd(dgamma=None, dbeta=None, workspace=...): ordered sums of its partial rows into the gradients."""
    check(load().nkb_layernorm_param_reduce(ptr(workspace), rows, D, planes, ptr(dgamma), ptr(dbeta), ptr(colsum), stream()),
          "layernorm_param_reduce")


def layernorm_ws(D):
    return load().nkb_layernorm_workspace_floats(D)


def gelu(dtype, x, dy, out, n):
    check(load().nkb_gelu(dtype, ptr(x), ptr(dy), ptr(out), n, stream()), "gelu")


def splitk_reduce(dtype, partial, splits, M, N, y, ldy, bias, stats):
    check(load().nkb_splitk_reduce(dtype, ptr(partial), splits, M, N, ptr(y), ldy, ptr(bias), ptr(stats), stream()), "splitk_reduce")


def wfold(dtype, w, scale, dst, Cout, K):
    check(load().nkb_wfold(dtype, ptr(w), ptr(scale), ptr(dst), Cout, K, stream()), "wfold")


def image_prep(src, sizes, flags, out, B, Hs, Ws, Ho, Wo, mean, std, fill=0.0):
    m, sd = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    check(load().nkb_image_prep(ptr(src), ptr(sizes), ptr(flags), ptr(out), B, Hs, Ws, Ho, Wo, C.cast(m, C.c_void_p),
                                C.cast(sd, C.c_void_p), float(fill), stream()), "image_prep")


def gelu_fwd_dgelu(dtype, x, y, gp, n):
    check(load().nkb_gelu_fwd_dgelu(dtype, ptr(x), ptr(y), ptr(gp), n, stream()), "gelu_fwd_dgelu")


def relu6(dtype, x, dy, out, n):
    check(load().nkb_relu6(dtype, ptr(x), ptr(dy), ptr(out), n, stream()), "relu6")


def scale_rows(dtype, x, add, out, scale, rows, inner):
    check(load().nkb_scale_rows(dtype, ptr(x), ptr(add), ptr(out), ptr(scale), rows, inner, stream()), "scale_rows")


def attn_softmax(dtype, backward, s, lds, p_in, out, ldp, rows, cols, scale):
    check(load().nkb_attn_softmax(dtype, int(backward), ptr(s), lds, ptr(p_in), ptr(out), ldp, rows, cols, scale, stream()),
          "attn_softmax")


def head_transpose(dtype, src, ld_in, sio, sii, outer, inner, out, T, dh, ldt):
    check(load().nkb_head_transpose(dtype, ptr(src), ld_in, sio, sii, outer, inner, ptr(out), T, dh, ldt, stream()),
          "head_transpose")


def vit_assemble(dtype, backward, tok, cls, pos, x, B, Tn, D):
    check(load().nkb_vit_assemble(dtype, int(backward), ptr(tok), ptr(cls), ptr(pos), ptr(x), B, Tn, D, stream()), "vit_assemble")


def colsum2d(dtype, x, out, rows, C_, ld, workspace=None):
    if workspace is not None and workspace.numel() < 256 * C_:
        raise RuntimeError("colsum2d: workspace needs 256 * C floats")
    check(load().nkb_colsum2d(dtype, ptr(x), ptr(out), rows, C_, ld, ptr(workspace), stream()), "colsum2d")


def dropout(dtype, backward, src, add, out, mask, n, p, seed=0):
    check(load().nkb_dropout(dtype, int(backward), ptr(src), ptr(add), ptr(out), ptr(mask), n, p, seed, stream()), "dropout")


def attn_forward(dtype, qkv, out, lse, B, T, H, dh, scale, outq=None, q_state=None):
    """outq / q_state: optional e4m3 copy of out for an fp8 projection (see fp8_quantize)."""
    check(load().nkb_attn_forward(dtype, ptr(qkv), ptr(out), ptr(lse), B, T, H, dh, scale, ptr(outq), ptr(q_state), stream()),
          "attn_forward")


def attn_backward(dtype, qkv, dout, out, lse, dqkv, B, T, H, dh, scale, dqkv_q=None, q_state=None, colsum=None, colsum_work=None):
    """dqkv_q / q_state: optional e5m2 copy of dqkv for the fp8 qkv gradients.  colsum / colsum_work ([B][3 H dh] scratch):
    colsum += the column sums of dqkv (the qkv bias gradient)."""
    check(load().nkb_attn_backward(dtype, ptr(qkv), ptr(dout), ptr(out), ptr(lse), ptr(dqkv), B, T, H, dh, scale, ptr(dqkv_q),
                                   ptr(q_state), ptr(colsum), ptr(colsum_work), stream()), "attn_backward")


def attn_backward_ds(dtype, qkv, dout, lse, P, dS, ldp, B, T, H, dh, scale, dq=None, ld_dq=0):
    check(load().nkb_attn_backward_ds(dtype, ptr(qkv), ptr(dout), ptr(lse), ptr(P), ptr(dS), ldp, B, T, H, dh, scale,
                                      ptr(dq), ld_dq, stream()), "attn_backward_ds")


def linear_gelu_fused_ok(dtype, M, K, N) -> bool:
    """act 5 of linear_gelu (y = gelu(pre), y2 = gelu'(pre) in the fc1 epilogue) is available for this shape."""
    return bool(load().nkb_linear_gelu_fused_ok(dtype, M, K, N))


def linear_residual_scaled(dtype, x, w, bias, add, row_scale, rows_per_sample, y, M, K, N):
    """y = add + row_scale[m // rows_per_sample] * (x @ w^T + bias) in one launch (shapes with linear_gelu_fused_ok)."""
    check(load().nkb_linear_residual_scaled(dtype, ptr(x), ptr(w), ptr(bias), ptr(add), ptr(row_scale), int(rows_per_sample),
                                            ptr(y), M, K, N, stream()), "linear_residual_scaled")


def linear_gelu(dtype, act, x, w, bias, aux, y, y2, M, K, N):
    check(load().nkb_linear_gelu(dtype, act, ptr(x), ptr(w), ptr(bias), ptr(aux), ptr(y), ptr(y2), M, K, N, stream()),
          "linear_gelu")
