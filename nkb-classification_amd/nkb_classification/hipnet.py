"""Layer-level executor over libnkbhip: forward / backward of conv+BN(+ReLU,+residual) stages, pooling and
the classifier head(s), with persistent NHWC activation buffers.  This is the host side of
`preds = model(img)` (engine.py:48) and of `loss.backward()` (engine.py:55-58) for ResNet-family backbones
(timm layout, /root/reference/nkb_classification/model.py:82).

Nothing here computes: every method only sizes buffers and enqueues libnkbhip kernels on the current stream.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
from torch import nn

from . import hip, runtime
from .runtime import ParamArena, Workspace, _round_up

_PACKED_STEM = os.environ.get("NKB_PACKED_STEM", "1") != "0"
# reduction pass of an interior BN stage's backward folded into the epilogue of the dgrad that produces its input
_FUSED_BN_BWD = os.environ.get("NKB_FUSED_BNBWD", "1") != "0"
# 3x3 stride-2 data gradients as four parity-class launches (9 taps instead of 36 multiplied, 27 of them by zero)
_S2_CLASSES = os.environ.get("NKB_S2_CLASSES", "1") != "0"
_FUSED_RES_BN_BWD = os.environ.get("NKB_FUSED_RES_BNBWD", "1") != "0"
# residual-closing stages keep a 1-bit/element ReLU mask; backward reads it instead of the activation and the masked
# block-output gradient is never materialised (consumers apply the bits on the fly)
_RELU_BITS = os.environ.get("NKB_RELU_BITS", "1") != "0"
_ATTN_FUSED_BWD = True  # whole attention backward in one kernel
_ATTN_FUSED_DQ = True   # dQ inside the attention backward-dS kernel
_SPLITK = True              # split-K for skinny Linear layers with K >= 32768
_EVAL_FOLD = os.environ.get("NKB_EVAL_FOLD", "1") != "0"     # eval mode: BatchNorm folded into the conv (one launch per stage)
# weight / bias gradients through per-split slabs + an ordered second stage instead of fp32 atomics: bit-identical across runs
_FP8_FUSED_QUANT = os.environ.get("NKB_FP8_FUSED_QUANT", "1") != "0"   # fp8 operands written by the producing kernel's epilogue
_FP8_WGRAD = True   # fp8 mode: weight gradients of the fp8 Linear layers on the fp8 kernel too
# bias gradients summed inside the e5m2 quantisation pass of dY (main stream) instead of a column-sum pass on the side stream:
# measured slower on unicom ViT-L/14 (65.0 vs 63.9 ms/step: the side stream has the slack, the main stream does not) — off
_FP8_COLSUM = False
_WPREP_FROM_SHADOW = True
# LayerNorm dgamma / dbeta reduction (two small launches) on the side stream instead of in the backward chain: measured neutral
# (unicom fp8 50.8-51.4 vs 50.9-51.3 ms, ViT-B/16 35.65-35.74 vs 35.74-35.78) — the other stream fills those gaps anyway.  Off.
_LN_REDUCE_SIDE = False
_LN_BWD_SCALED_COPY = True   # bf16: LayerNorm backward writes scale[b] * dx as well
_ROWSCALE_EPILOGUE = True   # bf16: drop-path scale in the residual GEMM epilogue
_GELU_EPILOGUE = True   # gelu + gelu' in the fc1 epilogue of the eight-phase core
_FP8_ATTN_COLSUM = True   # qkv bias gradient from the attention backward kernel's stores
_FP8_LN_BWD_QUANT = True   # LayerNorm backward writes the next Linear backward's fp8 operand
_FP8_EPI_COLSUM = True   # fc1's bias gradient from the fc2 data gradient's epilogue
_FP8_MASK_BITS = True   # fp8 step: ReLU6 output kept as fp8 operand + mask bits, no bf16 copy
_DET_WGRAD = os.environ.get("NKB_DET_WGRAD", "1") != "0"
# Gram form of the bottleneck closing stage (csrc/grambn.hip): BatchNorm statistics of conv3's output from the Gram matrix of its
# input, normalisation + shortcut + ReLU in conv3's epilogue, backward through R = g^T a and one K-concatenated data gradient — the
# raw conv output c3 and its gradient never exist in HBM (0 = the separate bn_apply / bn_backward passes, for A/B runs)
_GRAM_BN = os.environ.get("NKB_GRAM_BN", "1") != "0"
_GRAM_MAX_C = int(os.environ.get("NKB_GRAM_MAX_C", "128"))
# 3x3 / stride-1 forward and data gradient on the row-balanced DMA-pipelined core (csrc/convp.hip) where nkb_convp_tiles says eligible
_CONVP = os.environ.get("NKB_CONVP", "1") != "0"


class HipEngine:
    """One instance per (model, compute dtype).  dtype: torch.float32 (parity mode) or torch.bfloat16."""

    def __init__(self, arena: ParamArena, device, dtype: torch.dtype):
        self.arena = arena
        self.device = device
        self.T = dtype
        self.d = hip.dt(dtype)
        self.esz = 2 if dtype == torch.bfloat16 else 4
        self.kte = 128 // self.esz          # K elements per k-tile (Cin granularity of the GEMM kernels)
        self.ws = Workspace(device)
        self.saved: Dict[str, dict] = {}
        self._wver = -1
        self._wd: Dict[int, torch.Tensor] = {}     # id(conv/linear weight) -> dgrad-layout shadow
        self._wpad: Dict[int, torch.Tensor] = {}   # id(stem weight) -> K-padded forward shadow
        self._wd_cls: Dict[int, list] = {}         # id(3x3 stride-2 weight) -> four parity-class dgrad shadows
        self._convs: List[nn.Module] = []
        self._stems: List[nn.Module] = []
        self._heads = None
        self._bn_scratch_off = 0
        # weight gradients run on a side HIP stream next to the dgrad -> BN-backward chain of the main stream (both are
        # latency-bound at ~20 % MFMA busy, so they overlap almost additively); see begin_block()/on_side()
        self.overlap_wgrad = os.environ.get("NKB_WGRAD_STREAM", "1") != "0"
        self.fused_attention = os.environ.get("NKB_FUSED_ATTN", "1") != "0"
        self._side: Optional[torch.cuda.Stream] = None
        self._side_done: Dict[int, torch.cuda.Event] = {}
        self._suffix = ""
        self._fold: Dict[str, tuple] = {}          # stage key -> (fold_key, folded filter, shift) for eval mode
        self.fold_key = None                       # set by the owning classifier: (arena version, eval phase counter)
        # fp8 Linear contractions (BASELINE configs[4]): set by the owning classifier from cfg.amp_dtype == "fp8"
        self.fp8 = False
        self.fp8_candidates = None                 # Linear modules that run through linear() / linear_backward() (set by the model)
        self._f8w: Dict[int, tuple] = {}           # id(weight) -> (e4m3 [N][K], e4m3 dgrad layout [K][N], state)
        self._f8jobs = None                        # (all jobs, forward-layout jobs only) device tables for nkb_fp8_multi
        self._f8act: Dict[str, torch.Tensor] = {}  # activation / gradient site -> scaling state {scale, 1/scale, amax}
        self._f8kind: Dict[str, int] = {}
        self._f8table = None                       # device table of all site states for the one-launch scale update
        self._f8ready: Dict[str, torch.Tensor] = {}
        self._ones_cache: Dict[int, torch.Tensor] = {}
        self._gs_ready: Dict[int, torch.Tensor] = {}      # data_ptr of a gradient -> its stochastic-depth-scaled copy
        self._f8bias: Dict[str, bool] = {}          # ready sites whose producer also summed the columns (bias gradient done)
        # recorded launch plans of the train step (hip.Plan): key -> (plan, workspace generation, saved-activation table)
        self.plans: Dict[tuple, tuple] = {}
        self.plan_seen: Dict[tuple, int] = {}      # key -> workspace generation after its last eager run
        self.gram_bn = _GRAM_BN                    # Gram form of bottleneck closing stages (tests flip it per engine)
        self._gram_of = None                       # (data_ptr of an activation, its Gram matrix + column sums) from nkb_bn_apply_gram
        self._gram_ds_grad = None                  # input gradient of a K-concatenated projection shortcut (gram_closing_backward)

    # ------------------------------------------------------------------ weights ----
    def register(self, convs, stems, head_weights, head_biases):
        self._convs, self._stems = list(convs), list(stems)
        self._heads = (list(head_weights), list(head_biases))

    def kpad(self, k: int) -> int:
        return _round_up(k, self.kte)

    def refresh_weights(self, need_dgrad: bool):
        a = self.arena
        a.poll_external_writes()
        if self._wver == a.version and (not need_dgrad or self._dgrad_ready):
            return
        if self.T == torch.bfloat16:
            fresh = a.shadow is not None and getattr(a, "shadow_version", -1) == a.version     # written by the fused optimizer
            a.ensure_shadow()
            if not fresh:
                hip.wprep(self.d, a.flat_param, a.shadow, 1, 1, a.total, a.total, 0)
        for conv in self._stems:
            w = conv.weight
            if self.packed_stem(conv):
                buf = self._wpad.get(id(w))
                if buf is None:
                    buf = self._wpad[id(w)] = runtime.empty(w.shape[0], hip.stem_weight_cols(self.d), device=self.device,
                                                          dtype=self.T)
                hip.stem_wprep(self.d, a.param_flat(w), buf, w.shape[0], w.shape[1])
                continue
            K = w.shape[1] * w.shape[2] * w.shape[3]
            buf = self._wpad.get(id(w))
            if buf is None:
                buf = self._wpad[id(w)] = runtime.empty(w.shape[0], self.kpad(K), device=self.device, dtype=self.T)
            hip.wprep(self.d, a.param_flat(w), buf, w.shape[0], 1, K, buf.shape[1], 0)
        if need_dgrad:
            if self._wjobs is None:
                self._wjobs = self._build_dgrad_jobs()
            jobs, njobs, nblocks = self._wjobs
            # (bf16: the transposes read the optimizer's bf16 shadow of the masters — the values they would round to, half the bytes)
            hip.wprep_multi(self.d, a.flat_param, jobs, njobs, nblocks,
                            shadow=a.shadow if (self.T == torch.bfloat16 and _WPREP_FROM_SHADOW) else None)
        if self.fp8 and self.T == torch.bfloat16 and need_dgrad:
            self._refresh_fp8_weights()
        self._dgrad_ready = need_dgrad
        self._wver = a.version

    _dgrad_ready = False
    _wjobs = None

    def _build_dgrad_jobs(self):
        """Job table of nkb_wprep_multi for every dgrad-layout shadow ([Cin][R][S][Cout], the four parity classes of the
        3x3 stride-2 filters, the fused head); built once — parameter offsets and shadow buffers never move."""
        a = self.arena
        rows, nblocks = [], 0

        def add(src_off, dst, A, B, C, ld, mode):
            nonlocal nblocks
            rows.append([src_off, dst.data_ptr(), A, B, C, ld, mode, nblocks])
            nblocks += hip.wprep_job_blocks(A, B, C, ld, mode)

        for conv in self._convs:
            w = conv.weight
            co, ci, r, s = w.shape if w.dim() == 4 else (w.shape[0], w.shape[1], 1, 1)
            buf = self._wd[id(w)] = runtime.empty(ci, r, s, co, device=self.device, dtype=self.T)
            add(a.offset_of(w), buf, co, r * s, ci, co, 1)
            if self.s2_classes(conv):
                cls = self._wd_cls[id(w)] = [runtime.empty(ci, (2 if k >> 1 else 1) * (2 if k & 1 else 1), co,
                                                         device=self.device, dtype=self.T) for k in range(4)]
                for k in range(4):
                    add(a.offset_of(w), cls[k], co, 9, ci, co, 2 + k)
        hw, _ = self._heads
        ctot = sum(w.shape[0] for w in hw)
        E = hw[0].shape[1]
        cp = self.kpad(ctot)
        buf = self._wd["head"] = runtime.empty(E, cp, device=self.device, dtype=self.T)
        add(a.offset_of(hw[0]), buf, ctot, 1, E, cp, 1)
        jobs = torch.tensor(rows, dtype=torch.int64, device=self.device)
        return jobs, len(rows), nblocks

    # ------------------------------------------------------------------ fp8 ----
    @staticmethod
    def _fp8_shape_ok(K: int, N: int) -> bool:
        return K % 128 == 0 and K >= 256 and N % 256 == 0

    def _refresh_fp8_weights(self):
        """e4m3 copies of every eligible Linear weight in both operand layouts, with a just-in-time per-tensor scale (the
        amax of the freshly updated weights): four launches over a device job table, once per optimizer step."""
        a = self.arena
        if self._f8jobs is None:
            rows_all, rows_fwd, nb_all, nb_fwd = [], [], 0, 0
            for lin in (self.fp8_candidates if self.fp8_candidates is not None else self._convs):
                w = lin.weight
                if w.dim() != 2:
                    continue
                N, K = w.shape
                # forward x[M][K] . w[N][K]^T needs (K, N) eligible; the data gradient g[M][N] . wd[K][N]^T needs (N, K)
                if not (self._fp8_shape_ok(K, N) and self._fp8_shape_ok(N, K) and id(w) in self._wd):
                    continue
                st = torch.tensor([1.0, 1.0, 0.0], device=self.device)
                wq = runtime.empty(N, K, device=self.device, dtype=torch.uint8)
                wdq = runtime.empty(K, N, device=self.device, dtype=torch.uint8)
                self._f8w[id(w)] = (wq, wdq, st)
                n = N * K
                for src, dst, table in ((a.shadow_flat(w), wq, "both"), (self._wd[id(w)], wdq, "all")):
                    row = [src.data_ptr(), dst.data_ptr(), n, st.data_ptr(), hip.E4M3]
                    rows_all.append(row + [nb_all]); nb_all += hip.fp8_job_blocks(n)
                    if table == "both":
                        rows_fwd.append(row + [nb_fwd]); nb_fwd += hip.fp8_job_blocks(n)
            mk = lambda r: torch.tensor(r, dtype=torch.int64, device=self.device) if r else None    # noqa: E731
            self._f8jobs = (mk(rows_all), len(rows_all), nb_all, mk(rows_fwd), len(rows_fwd), nb_fwd)
        jall, nall, ball, jfwd, nfwd, bfwd = self._f8jobs
        if nall == 0:
            return
        hip.fp8_multi(3, jfwd, nfwd, 0)            # amax <- 0
        hip.fp8_multi(0, jfwd, nfwd, bfwd)         # amax of the current weights
        hip.fp8_multi(2, jfwd, nfwd, 0)            # scale <- 448 / amax
        hip.fp8_multi(1, jall, nall, ball)         # quantise both layouts

    def _fp8_operand(self, key: str, x: torch.Tensor, kind: int, colsum: Optional[torch.Tensor] = None, row_scale=None):
        """fp8 copy of an activation (e4m3) or gradient (e5m2) with delayed per-tensor scaling: the scale comes from the amax
        the previous step's pass over this site accumulated (the first call measures it just in time).  colsum: a [C] fp32
        vector that receives (+=) the column sums of x in the same pass — returns (q, state, colsum_done)."""
        q = self._f8ready.pop(key, None)
        if q is not None:                           # the producing kernel wrote it (and accumulated the amax) in its epilogue
            return q, self._f8act[key], self._f8bias.pop(key, False)
        st = self._fp8_state(key, kind, x)
        n = x.numel()
        q = self.ws.get(key + ".q", tuple(x.shape), torch.uint8)
        if (_FP8_COLSUM or row_scale is not None) and colsum is not None and self._fp8_colsum_ok(x):
            rows, C = x.shape
            work = self.ws.at_least("f8.colsum." + self._stream_tag(), hip.fp8_quantize_colsum_workspace(rows, C), torch.float32)
            hip.fp8_quantize_colsum(kind, x, rows, C, C, st, q, colsum, work,
                                    row_scale=row_scale[0] if row_scale else None, rows_per_sample=row_scale[1] if row_scale else 0)
            return q, st, True
        assert row_scale is None, "a row-scaled operand needs the fused pass (checked by the caller)"
        hip.fp8_quantize(self.d, kind, x, n, st, q)
        return q, st, False

    def _fp8_state(self, key: str, kind: int, x: Optional[torch.Tensor] = None):
        """Scaling state of a delayed-scaling site.  Existing sites are updated (scale from the previous step's amax, amax <- 0)
        all at once by fp8_begin_step(); a new site measures x just in time (or is None when there is nothing to measure yet)."""
        st = self._f8act.get(key)
        if st is None and x is not None:
            st = self._f8act[key] = torch.tensor([1.0, 1.0, 0.0], device=self.device)
            self._f8kind[key] = kind
            self._f8table = None
            hip.fp8_amax(self.d, x, x.numel(), st)
            hip.fp8_scale_update(st, kind)
        return st

    def fp8_begin_step(self):
        """One launch at the start of a forward pass: every activation / gradient site's scale from the amax its tensor showed
        in the previous step (192 single-thread launches per unicom ViT-L/14 step otherwise)."""
        self._f8ready.clear()                        # (an operand a producer wrote but nobody consumed does not outlive its step)
        self._f8bias.clear()
        self._gs_ready.clear()
        if not (self.fp8 and self._f8act):
            return
        if self._f8table is None:
            rows = [[0, 0, 0, st.data_ptr(), self._f8kind[k], 0] for k, st in self._f8act.items()]
            self._f8table = (torch.tensor(rows, dtype=torch.int64, device=self.device), len(rows))
        hip.fp8_multi(4, self._f8table[0], self._f8table[1], 0)

    def _fp8_colsum_ok(self, x: torch.Tensor) -> bool:
        return x.dim() == 2 and x.shape[1] % 512 == 0 and x.is_contiguous() and self.T == torch.bfloat16

    def _fp8_produce(self, key: Optional[str], shape, kind: int):
        """Second output of an fp8 GEMM: (buffer, state, kind) for the site `key` that will consume this tensor as its fp8
        operand, or None — on the first step, while the site has no measured scale yet, the consumer quantises by itself."""
        if key is None or not _FP8_FUSED_QUANT:
            return None
        st = self._f8act.get(key)
        if st is None:
            return None
        q = self.ws.get(key + ".q", tuple(shape), torch.uint8)
        self._f8ready[key] = q
        return q, st, kind

    def _fp8_linear_ok(self, lin, M: int) -> bool:
        return self.fp8 and self.T == torch.bfloat16 and id(lin.weight) in self._f8w

    @staticmethod
    def s2_classes(conv) -> bool:
        return (_S2_CLASSES and isinstance(conv, nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (2, 2)
                and conv.padding == (1, 1))

    @staticmethod
    def packed_stem(conv: nn.Conv2d) -> bool:
        """ResNet conv1 (<=4 -> Cout channels, 7x7, stride 2, pad 3) runs as an implicit GEMM on the packed NHWC image
        (nkb_stem_conv); any other stem goes through im2row.  NKB_PACKED_STEM=0 forces im2row (A/B measurements)."""
        return (_PACKED_STEM and conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3)
                and conv.in_channels <= 4 and conv.bias is None and conv.groups == 1)

    def w_fwd(self, w: torch.Tensor) -> torch.Tensor:
        if id(w) in self._wpad:
            return self._wpad[id(w)]
        return self.arena.shadow_flat(w) if self.T == torch.bfloat16 else self.arena.param_flat(w)

    # ------------------------------------------------------------------ forward ops ----
    def conv_bn(self, key: str, x: torch.Tensor, conv: nn.Conv2d, bn: nn.BatchNorm2d, relu: bool,
                res: Optional[torch.Tensor], train: bool, col_input: bool = False, pool: bool = False,
                stem_packed=None, defer_apply: bool = False, res_affine=None, gram: bool = False, gram_out: bool = False,
                proj=None):
        """y = act(bn(conv(x)) (+ res)).  x: [N,H,W,Cin] in the compute dtype (or the im2row matrix of the stem).
        stem_packed=(N, H, W): x is the packed image of nkb_stem_pack and conv the 7x7/2 stem.
        defer_apply=True: stop after the statistics and return (c, scale, shift) — for a projection shortcut, whose
        normalisation the consuming stage applies on the fly (res=c, res_affine=(scale, shift)).
        pool=True (stem): y = maxpool3x3s2(relu(bn(conv(x)))) in one pass over the raw conv output; the un-pooled
        activation is never materialised."""
        w = conv.weight
        if w.dim() == 2:
            # Linear -> BatchNorm1d (the unicom `feature` head): a 1x1 convolution over a [N,1,1,K] activation
            (co, ci), R, S, st, pad = w.shape, 1, 1, 1, 0
            x = x.view(x.shape[0], 1, 1, x.shape[-1])
        else:
            co, ci, R, S = w.shape
            st, pad = conv.stride[0], conv.padding[0]
        packed = stem_packed
        if packed:
            N, H, W = stem_packed
            P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            geom = dict(N=N, H=H, W=W, Cout=co, P=P, Q=Q)
        elif col_input:
            N, P, Q, kp = x.shape
            geom = dict(N=N * P * Q, H=1, W=1, Cin=kp, ldx=kp, P=1, Q=1, Cout=co, ldy=co, R=1, S=1, stride=1, pad=0)
        else:
            N, H, W, _ = x.shape
            P, Q = (H + 2 * pad - R) // st + 1, (W + 2 * pad - S) // st + 1
            geom = dict(N=N, H=H, W=W, Cin=ci, ldx=ci, P=P, Q=Q, Cout=co, ldy=co, R=R, S=S, stride=st, pad=pad)
        rows = N * P * Q
        if not train and _EVAL_FOLD and not packed and not col_input and not pool:
            # eval fast path: filter * scale (running statistics) once per eval phase, then conv + shift (+ res) (+ ReLU)
            # in one launch; nothing is saved and the raw conv output is never written
            sc = self.ws.get(key + ".bnvec", (4, co), torch.float32)
            ent = self._fold.get(key)
            if ent is None or ent[0] != self.fold_key:
                hip.bn_finalize(None, 0, co, rows, bn.weight, bn.bias, bn.running_mean, bn.running_var, 0.1, bn.eps, False,
                                sc[0], sc[1], sc[2], sc[3])
                wf = ent[1] if ent is not None else runtime.empty(co, R * S * ci, device=self.device, dtype=self.T)
                shift = ent[2] if ent is not None else runtime.empty(co, device=self.device, dtype=torch.float32)
                hip.wfold(self.d, self.arena.param_flat(w), sc[0], wf, co, R * S * ci)
                shift.copy_(sc[1])
                ent = self._fold[key] = (self.fold_key, wf, shift)
            y = self.ws.get(key + ".y", (N, P, Q, co), self.T)
            assert res_affine is None or res_affine[0] is None
            if w.dim() == 2 and self._splitk_ok(rows, ci, co) and res is None and not relu:
                S = 32                      # skinny Linear with a very long reduction: K-slices + one summing pass (see below)
                part = self.ws.get(key + ".splitk", (S, rows, co), torch.float32)
                hip.gemm_batched(self.d, x, ent[1], part, rows, co, ci // S, ci, ci, co, S, 1, (ci // S, 0), (ci // S, 0),
                                 (rows * co, 0), out_f32=True)
                hip.splitk_reduce(self.d, part, S, rows, co, y, co, ent[2], None)
            else:
                hip.conv_gemm(self.d, 0, x, ent[1], y, bias=ent[2], relu=relu, add=res, ldadd=co if res is not None else 0, **geom)
            return (y, None, None) if defer_apply else y
        if proj is not None:
            assert gram and train and self.gram_proj_ok(proj[0], conv)
            return self._conv_bn_gram(key, x, conv, bn, None, None, geom, rows, proj=proj)
        if gram and train and self.gram_ok(conv, res, x):
            return self._conv_bn_gram(key, x, conv, bn, res, res_affine, geom, rows)
        c = self.ws.get(key + ".c", (N, P, Q, co), self.T)
        bits = None
        tiles = hip.stat_tiles(self.d, rows, co)
        # 3x3 / stride 1 in bf16 (train mode): the row-balanced DMA-pipelined core (csrc/convp.hip), one partial-sum row per workgroup
        tiles_p = 0
        if train and _CONVP and not packed and not col_input and w.dim() == 4 and self.T == torch.bfloat16:
            tiles_p = hip.convp_tiles(self.d, 0, N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, ldy=co, R=R, S=S, stride=st, pad=pad)
        # 1x1 / stride 1 expansions (bottleneck conv3, Cout >= 2 Cin): the pixel-resident kernel (csrc/conv1p.hip)
        tiles_1 = 0
        if (train and _CONVP and not packed and not col_input and w.dim() == 4 and self.T == torch.bfloat16
                and R == 1 and S == 1 and st == 1 and pad == 0):
            tiles_1 = hip.conv1p_tiles(self.d, rows, ci, ci, co, co)
        tiles_s = hip.stemp_tiles(self.d, N, H, W, co) if (packed and train and _CONVP) else 0      # the stem through an LDS ring of image rows
        if tiles_p or tiles_1 or tiles_s:
            tiles = tiles_p or tiles_1 or tiles_s
        stats = self.ws.get(key + ".stats", (hip.bn_stats_floats(tiles, co),), torch.float32) if train else None
        if tiles_p:
            hip.convp_fwd(self.d, x, self.w_fwd(w), c, stats, N=N, H=H, W=W, Cin=ci, ldx=ci, Cout=co, ldy=co, tiles=tiles_p)
        elif tiles_1:
            hip.conv1p_fwd(self.d, x, self.w_fwd(w), c, stats, M=rows, Cin=ci, ldx=ci, Cout=co, ldy=co)
        elif tiles_s:
            hip.stemp_conv(self.d, x, self.w_fwd(w), c, stats, N, H, W, co, co)
        elif packed:
            hip.stem_conv(self.d, x, self.w_fwd(w), c, stats, N, H, W, co, co)
        elif w.dim() == 2 and self._splitk_ok(rows, ci, co):
            # skinny Linear with a very long reduction (unicom feature[0]: 128 x 262 144 -> 1 024 would be 8 workgroups of
            # 4 096 k-steps): 32 K-slices as one batched launch, then one pass that sums them and forms the BN statistics
            S = 32
            part = self.ws.get(key + ".splitk", (S, rows, co), torch.float32)
            hip.gemm_batched(self.d, x, self.w_fwd(w), part, rows, co, ci // S, ci, ci, co, S, 1, (ci // S, 0), (ci // S, 0),
                             (rows * co, 0), out_f32=True)
            hip.splitk_reduce(self.d, part, S, rows, co, c, co, None, stats)
        else:
            hip.conv_gemm(self.d, 0, x, self.w_fwd(w), c, stats=stats, **geom)
        sc = self.ws.get(key + ".bnvec", (4, co), torch.float32)
        scale, shift, mean, invstd = sc[0], sc[1], sc[2], sc[3]
        hip.bn_finalize(stats, tiles, co, rows, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                        bn.momentum if bn.momentum is not None else 0.1, bn.eps, train, scale, shift, mean, invstd)
        if defer_apply:
            assert not relu and res is None and not pool
            if train:
                self.saved[key] = dict(x=x, c=c, y=None, mean=mean, invstd=invstd, relu=False, geom=geom, conv=conv, bn=bn,
                                       rows=rows, col_input=col_input, scale=scale, shift=shift, has_res=False,
                                       pool_idx=None, stem_packed=bool(packed), bits=None)
            return c, scale, shift
        if pool:
            assert relu and res is None
            P2, Q2 = (P - 1) // 2 + 1, (Q - 1) // 2 + 1
            y = self.ws.get(key + ".y", (N, P2, Q2, co), self.T)
            idx = self.ws.get(key + ".idx", (N, P2, Q2, co), torch.uint8)
            # train: the raw value behind every pooled winner is kept for the backward reduction (one 16-byte read per pooled chunk there
            # instead of eight 2-byte gathers from c)
            xsel = self.ws.get(key + ".xsel", (N, P2, Q2, co), self.T) if train else None
            hip.bn_relu_maxpool(self.d, False, c, scale, shift, mean, invstd, None, y, idx, None, None, None, None,
                                N, P, Q, co, xsel=xsel)
        else:
            idx = None
            y = self.ws.get(key + ".y", (N, P, Q, co), self.T)
            # stages that close a residual block keep their ReLU mask as one bit per element for backward (the mask
            # cannot be recomputed from c alone there); y itself is then only read by the next block's convolutions
            if train and relu and res is not None and _RELU_BITS:
                bits = self.ws.get(key + ".bits", (rows, co // (8 if self.T == torch.bfloat16 else 4)), torch.uint8)
            if (gram_out and train and relu and res is None and co in (64, 128) and self.T == torch.bfloat16 and self.gram_bn):
                # the stage before a Gram-form closing stage: the same normalisation pass also leaves y^T y and the column sums of y
                gs = self.ws.get(key + ".gramout", (co * co + co,), torch.float32)
                work = self.ws.at_least("gram.slabs", hip.bn_apply_gram_ws(rows, co), torch.float32)
                hip.bn_apply_gram(self.d, c, y, scale, shift, rows, co, gs, work)
                self._gram_of = (y.data_ptr(), gs)
            else:
                hip.bn_apply(self.d, c, res, y, scale, shift, rows, co, relu, bits,
                             res_scale=res_affine[0] if res_affine else None, res_shift=res_affine[1] if res_affine else None)
        if train:
            self.saved[key] = dict(x=x, c=c, y=y, mean=mean, invstd=invstd, relu=relu, geom=geom, conv=conv, bn=bn,
                                   rows=rows, col_input=col_input, scale=scale, shift=shift, has_res=res is not None,
                                   pool_idx=idx, pool_xsel=xsel if pool else None, stem_packed=bool(packed), bits=bits)
        return y

    def gram_ok(self, conv, res, x) -> bool:
        """The Gram form exists for bf16 1x1 / stride-1 closing stages with 64 | Cin <= 512 and Cout > 64 (every timm Bottleneck conv3);
        it is TAKEN up to Cin = 128 (_GRAM_MAX_C; ResNet-50 layer1 / layer2, 7 of 16 blocks, 73 % of the closing-stage bytes): the small
        algebra costs O(Cout * Cin^2) whatever the image count, and the launches around it are latency-bound — same box, alternating:
        Cin <= 512 19.13 ms, <= 256 18.65 / 18.70, <= 128 18.40 (layer3: 14 x 14 maps, 103 MB of conv output per block; layer4: 51 MB)."""
        w = conv.weight
        return (self.gram_bn and _RELU_BITS and _FUSED_BN_BWD and _FUSED_RES_BN_BWD and self.T == torch.bfloat16 and res is not None
                and w.dim() == 4 and w.shape[2] == 1 and w.shape[3] == 1 and conv.stride == (1, 1) and conv.padding == (0, 0)
                and w.shape[1] % 64 == 0 and w.shape[1] <= _GRAM_MAX_C and w.shape[0] > 64 and w.shape[0] % 8 == 0 and x.dim() == 4)

    def gram_proj_ok(self, dconv, conv) -> bool:
        """The projection shortcut can ride inside the Gram-form closing convolution (K-concatenated): 1x1 / stride 1 / no bias on a
        block input of 64 | Cin <= _GRAM_MAX_C channels (timm ResNet-50: layer1.0) and an eligible closing convolution."""
        wd, w = dconv.weight, conv.weight
        return (self.gram_bn and self.T == torch.bfloat16 and wd.dim() == 4 and wd.shape[2] == 1 and wd.shape[3] == 1
                and dconv.stride == (1, 1) and dconv.padding == (0, 0) and dconv.bias is None and wd.shape[1] % 64 == 0
                and wd.shape[1] <= _GRAM_MAX_C and wd.shape[0] == w.shape[0]
                and self.gram_ok(conv, wd, torch.empty(0, 0, 0, 0)))

    def _gram_of_input(self, key, xin, rows, cx):
        """G = x^T x and the column sums of a NON-NEGATIVE activation (a block input): nkb_bn_apply_gram as an identity pass (scale 1,
        shift 0, no store) for 64 / 128 channels, the weight-gradient GEMM otherwise."""
        gs = self.ws.get(key + ".gramx", (cx * cx + cx,), torch.float32)
        if cx in (64, 128):
            one = self._ones_cache.get(("f", cx))
            if one is None:
                one = self._ones_cache[("f", cx)] = (torch.ones(cx, device=self.device), torch.zeros(cx, device=self.device))
            work = self.ws.at_least("gram.slabs", hip.bn_apply_gram_ws(rows, cx), torch.float32)
            hip.bn_apply_gram(self.d, xin, None, one[0], one[1], rows, cx, gs, work)
        else:
            N, H, W, _ = xin.shape
            self.wgrad(xin, xin, gs[:cx * cx], dbias=gs[cx * cx:], assign=True, N=N, H=H, W=W, Cin=cx, ldx=cx, P=H, Q=W, Cout=cx, lddy=cx)
        return gs

    def _conv_bn_gram(self, key, x, conv, bn, res, res_affine, geom, rows, proj=None):
        """Closing stage in the Gram form (train mode): G = x^T x and the column sums of x (one pass over the NARROW input), the
        batch statistics of conv(x) from them (nkb_gram_bn_stats), then y = relu(conv(x) * scale + shift + res) in ONE launch."""
        w = conv.weight
        co, ci = w.shape[0], w.shape[1]
        N, P, Q = geom["N"], geom["P"], geom["Q"]
        ready = self._gram_of
        self._gram_of = None
        if ready is not None and ready[0] == x.data_ptr():
            gs = ready[1]                           # left by the pass that produced x (nkb_bn_apply_gram)
            G, s = gs[:ci * ci], gs[ci * ci:]
        else:
            gs = self.ws.get(key + ".gram", (ci * ci + ci,), torch.float32)
            G, s = gs[:ci * ci], gs[ci * ci:]
            self.wgrad(x, x, G, dbias=s, assign=True, N=N, H=P, W=Q, Cin=ci, ldx=ci, P=P, Q=Q, Cout=ci, lddy=ci)
        sc = self.ws.get(key + ".bnvec", (4, co), torch.float32)
        scale, shift, mean, invstd = sc[0], sc[1], sc[2], sc[3]
        cov = self.ws.at_least("gram.cov", max(ci, proj[0].weight.shape[1] if proj is not None else 0) ** 2, torch.float32)
        mu = self.ws.get(key + ".gmu", (ci,), torch.float32)
        T = self.ws.get(key + ".gT", (co, ci), torch.float32)
        hip.gram_bn_stats(self.d, self.w_fwd(w), G, s, rows, ci, co, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                          bn.momentum if bn.momentum is not None else 0.1, bn.eps, cov, mu, T, scale, shift, mean, invstd)
        y = self.ws.get(key + ".y", (N, P, Q, co), self.T)
        bits = self.ws.get(key + ".bits", (rows, co // 8), torch.uint8)
        gram_ds = None
        if proj is not None:
            # projection shortcut inside the same launch: its BatchNorm statistics from the Gram matrix of the block input, both
            # scales folded into one K-concatenated filter — the shortcut's raw conv output never exists either
            dconv, dbn, xin, dkey = proj
            cx = dconv.weight.shape[1]
            gx = self._gram_of_input(dkey, xin, rows, cx)
            scd = self.ws.get(dkey + ".bnvec", (4, co), torch.float32)
            mud = self.ws.get(dkey + ".gmu", (cx,), torch.float32)
            Td = self.ws.get(dkey + ".gT", (co, cx), torch.float32)
            hip.gram_bn_stats(self.d, self.w_fwd(dconv.weight), gx[:cx * cx], gx[cx * cx:], rows, cx, co, dbn.weight, dbn.bias,
                              dbn.running_mean, dbn.running_var, dbn.momentum if dbn.momentum is not None else 0.1, dbn.eps, cov, mud, Td,
                              scd[0], scd[1], scd[2], scd[3])
            wf = self.ws.get(key + ".gwf", (co, ci + cx), self.T)
            shf = self.ws.get(key + ".gshift", (co,), torch.float32)
            hip.gram_fold2(self.d, self.w_fwd(w), scale, ci, self.w_fwd(dconv.weight), scd[0], cx, co, wf, shift, scd[1], shf)
            hip.conv_cat_relu_bits(self.d, x, ci, ci, xin, cx, cx, wf, shf, y, bits, rows, co, co)
            gram_ds = dict(conv=dconv, bn=dbn, x=xin, T=Td, mu=mud, mean=scd[2], invstd=scd[3], key=dkey)
        else:
            hip.conv_affine_residual(self.d, x, self.w_fwd(w), y, scale, shift, res, co, res_affine[0] if res_affine else None,
                                     res_affine[1] if res_affine else None, bits, **geom)
        self.saved[key] = dict(x=x, c=None, cshape=(N, P, Q, co), y=y, mean=mean, invstd=invstd, relu=True, geom=geom, conv=conv,
                               bn=bn, rows=rows, col_input=False, scale=scale, shift=shift, has_res=True, pool_idx=None,
                               stem_packed=False, bits=bits, gram=dict(T=T, mu=mu), gram_ds=gram_ds)
        return y

    def _gram_r(self, g, x, R, rows, co, ci, geom):
        """R[co][ci] = g^T x over the stage's pixels, on the MAIN stream (the coefficients of the block's data gradient wait for it);
        wgrad() takes the streaming kernel of csrc/gramr.hip where the shape is eligible, else the generic weight-gradient kernel."""
        self.wgrad(g, x, R, assign=True, N=geom["N"], H=geom["P"], W=geom["Q"], Cin=ci, ldx=ci, P=geom["P"], Q=geom["Q"], Cout=co, lddy=co)

    def gram_closing_backward(self, key: str, g: torch.Tensor, g_stats, prev_key: str, slot: str):
        """Backward of a Gram-form closing stage.  g: masked gradient of the block output with its per-tile sums (left by the next
        block's conv1 data gradient, nkb_conv_dgrad_bn).  R = g^T a on the main stream (its row dots with W are sum g c, which the
        coefficients need), then the small algebra (nkb_gram_bn_backward: dgamma, dbeta, dW, the concatenated filter) and ONE data
        gradient over [g | a] with the fused BN-backward epilogue of the stage before.  Returns (masked gradient, (stats, tiles)) for
        bn_backward_fused(prev_key, ...) exactly like conv_backward(..., fuse_bn=prev_key)."""
        sv, svp = self.saved[key], self.saved[prev_key]
        x, conv, bn, rows, geom = sv["x"], sv["conv"], sv["bn"], sv["rows"], sv["geom"]
        w = conv.weight
        co, ci = w.shape[0], w.shape[1]
        a = self.arena
        stats, tiles = g_stats
        R = self.ws.get(key + ".gR", (co, ci), torch.float32)
        cbias = self.ws.get(key + ".gcbias", (ci,), torch.float32)
        tiles2 = hip.stat_tiles(self.d, rows, ci)
        stats2 = self.ws.get(prev_key + ".bstats", (hip.bn_stats_floats(tiles2, ci),), torch.float32)
        self._gram_r(g, x, R, rows, co, ci, geom)
        wcat = self.ws.get(key + ".gwcat", (ci, co + ci), self.T)
        need = hip.gram_bn_backward_ws(ci, co)
        if sv.get("gram_ds") is not None:
            need = max(need, hip.gram_bn_backward_ws(sv["gram_ds"]["conv"].weight.shape[1], co))
        coef = self.ws.at_least("gram.bwd", need, torch.float32)
        hip.gram_bn_backward(self.d, self.w_fwd(w), R, sv["gram"]["T"], sv["gram"]["mu"], stats, tiles, rows, ci, co, bn.weight,
                             sv["mean"], sv["invstd"], a.grad_flat(bn.weight), a.grad_flat(bn.bias), a.grad_flat(w), wcat, cbias, coef)
        ds = sv.get("gram_ds")
        if ds is not None:
            # the K-concatenated projection shortcut: same g, the block input in place of a
            dconv, dbn, xin = ds["conv"], ds["bn"], ds["x"]
            cx = dconv.weight.shape[1]
            Rd = self.ws.get(ds["key"] + ".gR", (co, cx), torch.float32)
            self._gram_r(g, xin, Rd, rows, co, cx, geom)
            wcd = self.ws.get(ds["key"] + ".gwcat", (cx, co + cx), self.T)
            cbd = self.ws.get(ds["key"] + ".gcbias", (cx,), torch.float32)
            hip.gram_bn_backward(self.d, self.w_fwd(dconv.weight), Rd, ds["T"], ds["mu"], stats, tiles, rows, cx, co, dbn.weight,
                                 ds["mean"], ds["invstd"], a.grad_flat(dbn.weight), a.grad_flat(dbn.bias), a.grad_flat(dconv.weight), wcd, cbd, coef)
            dxs = self.scratch("t6", xin.shape)
            hip.conv_cat_bias(self.d, g, co, co, xin, cx, cx, wcd, cbd, dxs, rows, cx, cx)
            self._gram_ds_grad = dxs
        dx = self.scratch(slot, x.shape)
        hip.conv_dgrad_bn_cat(self.d, g, co, co, x, ci, ci, wcat, cbias, dx, svp["c"], svp["scale"], svp["shift"], svp["mean"],
                              stats2, rows, ci, ci)
        return dx, (stats2, tiles2)

    def _splitk_ok(self, rows: int, ci: int, co: int) -> bool:
        return _SPLITK and rows <= 256 and ci >= 32768 and co > 64 and ci % (self.kte * 32) == 0

    def maxpool(self, key: str, x: torch.Tensor, train: bool) -> torch.Tensor:
        N, H, W, C = x.shape
        P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = self.ws.get(key + ".y", (N, P, Q, C), self.T)
        idx = self.ws.get(key + ".idx", (N, P, Q, C), torch.uint8)
        hip.maxpool(self.d, False, x, y, idx, N, H, W, C)
        if train:
            self.saved[key] = dict(idx=idx, in_shape=(N, H, W, C))
        return y

    def avgpool(self, key: str, x: torch.Tensor) -> torch.Tensor:
        N, H, W, C = x.shape
        y = self.ws.get(key + ".y", (N, C), self.T)
        hip.avgpool(self.d, False, x, y, N, H * W, C)
        self.saved[key] = dict(in_shape=(N, H, W, C))
        return y

    def head(self, emb: torch.Tensor, train: bool, drop_p: float = 0.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Fused classifier heads: logits[B][sum C_t] (fp32) = emb @ W_all^T + b_all.
        With classifier dropout active (train, p > 0) every head draws its own mask (model.py:104-108 gives each head
        its own nn.Dropout), so the heads run as separate GEMMs on separately dropped copies of the embedding."""
        hw, hb = self._heads
        ctot = sum(w.shape[0] for w in hw)
        B, E = emb.shape
        a = self.arena
        if train and drop_p > 0:
            logits = out if out is not None else torch.empty(B, ctot, device=self.device, dtype=torch.float32)
            dropped, masks, lo_c = [], [], 0
            for t, w in enumerate(hw):
                d_emb = self.ws.get(f"head.drop{t}", (B, E), self.T)
                mask = self.ws.get(f"head.mask{t}", (B, E), torch.uint8)
                seed = hip.fresh_seed()
                hip.dropout(self.d, False, emb, None, d_emb, mask, B * E, drop_p, seed)
                n_t = w.shape[0]
                hip.conv_gemm(self.d, 0, d_emb, self.w_fwd(w), logits[:, lo_c:], N=B, H=1, W=1, Cin=E, ldx=E, P=1, Q=1,
                              Cout=n_t, ldy=ctot, bias=a.param_flat(hb[t]), out_f32=True)
                dropped.append(d_emb); masks.append(mask); lo_c += n_t
            self.saved["head"] = dict(emb=emb, ctot=ctot, dropped=dropped, masks=masks, drop_p=drop_p)
            return logits
        lo = a.offset_of(hw[0])
        wall = (a.shadow if self.T == torch.bfloat16 else a.flat_param)[lo:lo + ctot * E]
        bo = a.offset_of(hb[0])
        ball = a.flat_param[bo:bo + ctot]
        logits = out if out is not None else torch.empty(B, ctot, device=self.device, dtype=torch.float32)
        hip.conv_gemm(self.d, 0, emb, wall, logits, N=B, H=1, W=1, Cin=E, ldx=E, P=1, Q=1, Cout=ctot, ldy=ctot,
                      bias=ball, out_f32=True)
        if train:
            self.saved["head"] = dict(emb=emb, ctot=ctot)
        return logits

    # ------------------------------------------------------------------ backward ops ----
    def _stream_tag(self) -> str:
        # launches of one stream run in order, so one scratch buffer per stream is race-free
        return "side" if self._side is not None and torch.cuda.current_stream() == self._side else "main"

    def wgrad(self, dy, x, dw, *, dbias=None, assign=False, **geom):
        """nkb_conv_wgrad into the gradient arena, deterministic (slabs + ordered reduce) unless NKB_DET_WGRAD=0.
        assign: dw (a scratch product, not the arena) is overwritten — no memset in front of the launch."""
        if assign and not _DET_WGRAD:
            hip.zero_(dw)
            if dbias is not None:
                hip.zero_(dbias)
            assign = False
        # 1x1 products between a wide and a narrow stage (256 <-> 64, 512 <-> 128 channels) on long pixel ranges: the streaming kernel
        # of csrc/gramr.hip (the wide operand plays g; a convolution whose OUTPUT is the narrow one gets its gradient transposed) — for the
        # scratch products on the MAIN stream only (assign): as a side-stream weight gradient its one workgroup per CU at the full HBM
        # rate starves the main queue (measured: the six layer1 / layer2 launches give back the 0.14 ms the main-stream ones gain)
        if (assign and _CONVP and _DET_WGRAD and dbias is None and self.T == torch.bfloat16 and geom.get("R", 1) == 1 and geom.get("S", 1) == 1
                and geom.get("stride", 1) == 1 and geom.get("pad", 0) == 0 and geom["ldx"] == geom["Cin"] and geom["lddy"] == geom["Cout"]
                and geom["H"] == geom["P"] and geom["W"] == geom["Q"]):
            rows, ci, co = geom["N"] * geom["P"] * geom["Q"], geom["Cin"], geom["Cout"]
            wide_out = co > ci
            need = hip.gramr_workspace(self.d, rows, max(co, ci), min(co, ci))
            if need:
                work = self.ws.at_least("wgrad.slabs." + self._stream_tag(), need, torch.float32)
                if wide_out:
                    hip.gramr(self.d, dy, co, x, ci, dw, rows, co, ci, work, assign=assign)
                else:
                    hip.gramr(self.d, x, ci, dy, co, dw, rows, ci, co, work, assign=assign, transposed=True)
                return
        work = None
        if _DET_WGRAD:
            need = hip.conv_wgrad_workspace(self.d, N=geom["N"], P=geom["P"], Q=geom["Q"], Cin=geom["Cin"], Cout=geom["Cout"],
                                            R=geom.get("R", 1), S=geom.get("S", 1), stride=geom.get("stride", 1),
                                            pad=geom.get("pad", 0), has_bias=dbias is not None)
            work = self.ws.at_least("wgrad.slabs." + self._stream_tag(), need, torch.float32)
        hip.conv_wgrad(self.d, dy, x, dw, dbias=dbias, workspace=work, assign=assign, **geom)

    def colsum2d(self, x, out, rows, C_, ld):
        """Column sums over many rows (bias / position-embedding gradients), ordered two-stage sum when rows span blocks."""
        if _DET_WGRAD and self.T == torch.bfloat16 and C_ % 512 == 0 and ld % 8 == 0 and rows >= 1024:
            # wide bf16 matrices: the 16-bytes-per-thread column-sum pass (the fp8 path's kernel without its quantisation)
            work = self.ws.at_least("colsum.part." + self._stream_tag(), hip.fp8_quantize_colsum_workspace(rows, C_), torch.float32)
            hip.fp8_quantize_colsum(0, x, rows, C_, ld, None, None, out, work)
            return
        work = None
        if _DET_WGRAD and rows > 256:
            work = self.ws.at_least("colsum.part." + self._stream_tag(), 256 * C_, torch.float32)
        hip.colsum2d(self.d, x, out, rows, C_, ld, workspace=work)

    def scratch(self, slot: str, shape) -> torch.Tensor:
        return self.ws.get("grad.%s%s.%s" % (slot, self._suffix, "x".join(str(int(v)) for v in shape)), shape, self.T)

    # ---- side stream for weight gradients ---------------------------------------------------------------------
    def on_side(self, fn):
        """Enqueue fn's kernels on the side stream, ordered after everything enqueued so far on the main stream."""
        if not self.overlap_wgrad:
            fn()
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        ev, side = torch.cuda.Event(), self._side
        hip.event_record(ev, torch.cuda.current_stream())
        hip.stream_wait_event(side, ev)
        with torch.cuda.stream(side):
            fn()

    def join_side(self):
        """Main stream waits for everything enqueued on the side stream so far (forward-pass use; the backward pass
        tracks its weight gradients per block with begin_block / end_block instead)."""
        if self._side is not None and self.overlap_wgrad:
            hip.stream_wait_stream(torch.cuda.current_stream(), self._side)

    def begin_block(self, index: int):
        """Scratch gradient buffers alternate between two sets by block parity; before a set is reused the main
        stream waits for the side-stream weight gradients that still read it (issued two blocks earlier)."""
        self._suffix = ".p%d" % (index & 1)
        ev = self._side_done.pop(index + 2, None)
        if ev is not None:
            hip.stream_wait_event(torch.cuda.current_stream(), ev)

    def end_block(self, index: int):
        if self._side is not None:
            ev = torch.cuda.Event()
            hip.event_record(ev, self._side)
            self._side_done[index] = ev

    def side_event(self):
        """Event marking the current tail of the weight-gradient stream (None while that stream is unused)."""
        if self._side is None or not self.overlap_wgrad:
            return None
        ev = torch.cuda.Event()
        ev.record(self._side)
        return ev

    def wait_side(self):
        """Main stream waits for every outstanding weight gradient (before the optimizer / the gradient exchange)."""
        if self._side is not None:
            hip.stream_wait_stream(torch.cuda.current_stream(), self._side)
        self._side_done.clear()
        self._suffix = ""

    def head_backward(self, glogits: torch.Tensor, need_demb: bool) -> Optional[torch.Tensor]:
        sv = self.saved["head"]
        emb, ctot = sv["emb"], sv["ctot"]
        hw, hb = self._heads
        B, E = emb.shape
        cp = self.kpad(ctot)
        a = self.arena
        if "dropped" in sv:
            return self._head_backward_dropout(glogits, need_demb)
        dl = self.ws.get("head.dl", (B, cp), self.T)
        if not glogits.is_contiguous():
            glogits = glogits.contiguous()
        hip.pad_cast(self.d, glogits, dl, B, ctot, ctot, cp)
        lo = a.offset_of(hw[0])
        self.wgrad(dl, emb, a.flat_grad[lo:lo + ctot * E], N=B, H=1, W=1, Cin=E, ldx=E, P=1, Q=1,
                       Cout=ctot, lddy=cp)
        bo = a.offset_of(hb[0])
        hip.colsum(self.d, dl, a.flat_grad[bo:bo + ctot], B, ctot, cp)
        if not need_demb:
            return None
        g = self.ws.get("head.demb", (B, E), self.T)
        hip.conv_gemm(self.d, 0, dl, self._wd["head"], g, N=B, H=1, W=1, Cin=cp, ldx=cp, P=1, Q=1, Cout=E, ldy=E)
        return g

    def _head_backward_dropout(self, glogits: torch.Tensor, need_demb: bool) -> Optional[torch.Tensor]:
        sv = self.saved["head"]
        hw, hb = self._heads
        emb, ctot, p = sv["emb"], sv["ctot"], sv["drop_p"]
        B, E = emb.shape
        a = self.arena
        if not glogits.is_contiguous():
            glogits = glogits.contiguous()
        g_total = self.ws.get("head.demb", (B, E), self.T) if need_demb else None
        lo_c = 0
        wd_all = self._wd.get("head")            # [E][cp] transposed, all heads
        cp = self.kpad(ctot)
        for t, w in enumerate(hw):
            n_t = w.shape[0]
            cpt = self.kpad(n_t)
            dl = self.ws.get(f"head.dl{t}", (B, cpt), self.T)
            hip.pad_cast(self.d, glogits[:, lo_c:], dl, B, n_t, ctot, cpt)
            self.wgrad(dl, sv["dropped"][t], a.grad_flat(w), N=B, H=1, W=1, Cin=E, ldx=E, P=1, Q=1, Cout=n_t,
                           lddy=cpt)
            hip.colsum(self.d, dl, a.grad_flat(hb[t]), B, n_t, cpt)
            if need_demb:
                wt = self.ws.get(f"head.wt{t}", (E, cpt), self.T)
                hip.wprep(self.d, a.param_flat(w), wt, n_t, 1, E, cpt, 1)
                gt = self.ws.get(f"head.demb{t}", (B, E), self.T)
                hip.conv_gemm(self.d, 0, dl, wt, gt, N=B, H=1, W=1, Cin=cpt, ldx=cpt, P=1, Q=1, Cout=E, ldy=E)
                # through the head's own dropout mask, accumulated over heads
                hip.dropout(self.d, True, gt, g_total if t > 0 else None, g_total, sv["masks"][t], B * E, p, 0)
            lo_c += n_t
        return g_total

    def avgpool_backward(self, key: str, g: torch.Tensor, slot: str) -> torch.Tensor:
        N, H, W, C = self.saved[key]["in_shape"]
        dx = self.scratch(slot, (N, H, W, C))
        hip.avgpool(self.d, True, g, dx, N, H * W, C)
        return dx

    def maxpool_backward(self, key: str, g: torch.Tensor, slot: str) -> torch.Tensor:
        sv = self.saved[key]
        N, H, W, C = sv["in_shape"]
        dx = self.scratch(slot, (N, H, W, C))
        hip.maxpool(self.d, True, g, dx, sv["idx"], N, H, W, C)
        return dx

    def bn_backward(self, key: str, g_y: torch.Tensor, slot: str, write_masked: bool = False,
                    g_bits: Optional[torch.Tensor] = None) -> torch.Tensor:
        """g_y: gradient w.r.t. the stage output y.  Returns the gradient w.r.t. the raw conv output.
        With write_masked the ReLU-masked g_y is written back in place (it then is the gradient that flows
        into the residual branch) — unless the stage kept a bit mask (saved["bits"]): then g_y stays as it is and every
        consumer applies the bits itself.  g_bits: bit mask to apply to g_y on the way in (the shortcut's BN backward,
        whose incoming gradient is the block-output gradient under the main branch's ReLU)."""
        sv = self.saved[key]
        bn, rows = sv["bn"], sv["rows"]
        co = sv["c"].shape[-1]
        a = self.arena
        gc = self.scratch(slot, sv["c"].shape)
        work = self.ws.at_least("bn.work", hip.bn_backward_ws(rows, co), torch.float32)
        # ReLU mask: stages without a residual recompute it from the raw conv output (one tensor read less)
        bits = sv.get("bits") if g_bits is None else g_bits
        from_y = sv["relu"] and sv["has_res"] and bits is None
        from_x = sv["relu"] and not sv["has_res"] and bits is None
        # with a bit mask nothing needs to be written back: consumers of g_y apply the same bits themselves
        masked_out = g_y if (write_masked and sv["relu"] and bits is None) else None
        hip.bn_backward(self.d, g_y, sv["c"], sv["y"] if from_y else None, sv["mean"], sv["invstd"], bn.weight, rows,
                        co, a.grad_flat(bn.weight), a.grad_flat(bn.bias), gc, masked_out,
                        work, fscale=sv["scale"] if from_x else None, fshift=sv["shift"] if from_x else None,
                        relu_bits=bits)
        return gc

    def bn_pool_backward(self, key: str, g_p: torch.Tensor, slot: str) -> torch.Tensor:
        """Backward of a pool=True stage: pooled gradient -> gradient w.r.t. the raw conv output (max-pool routing, ReLU
        mask and BatchNorm backward fused; bn weight/bias gradients accumulate into the arena)."""
        sv = self.saved[key]
        bn = sv["bn"]
        N, H, W, co = sv["c"].shape
        a = self.arena
        gc = self.scratch(slot, sv["c"].shape)
        work = self.ws.at_least("bn.work", hip.bn_relu_maxpool_ws(N, H, W, co), torch.float32)
        hip.bn_relu_maxpool(self.d, True, sv["c"], sv["scale"], sv["shift"], sv["mean"], sv["invstd"], bn.weight, g_p,
                            sv["pool_idx"], gc, a.grad_flat(bn.weight), a.grad_flat(bn.bias), work, N, H, W, co, xsel=sv.get("pool_xsel"))
        return gc

    def can_fuse_bn_backward(self, bn_key: str) -> bool:
        sv = self.saved[bn_key]
        return (_FUSED_BN_BWD and sv["relu"] and not sv["has_res"] and sv["pool_idx"] is None
                and sv["c"].shape[-1] % 8 == 0)

    def can_fuse_residual_bn_backward(self, bn_key: str, consumer_key: str) -> bool:
        """The stage `bn_key` closes a residual block (kept ReLU bits) and `consumer_key` is the first conv of the next
        block, whose data gradient (+ shortcut gradient) IS the gradient of that stage's output."""
        sv, cv = self.saved[bn_key], self.saved[consumer_key]
        return (_FUSED_BN_BWD and _FUSED_RES_BN_BWD and sv.get("bits") is not None and sv["y"].shape[-1] % 8 == 0
                and not self.s2_classes(cv["conv"]) and not cv["col_input"] and not cv["stem_packed"])

    def bn_backward_fused(self, key: str, g_masked: torch.Tensor, stats, slot: str) -> torch.Tensor:
        """Second half of conv_backward(..., fuse_bn=key): g_masked is already ReLU-masked and `stats` = (buffer, tiles)
        holds the per-tile sums, so only the finalize and the elementwise pass remain."""
        sv = self.saved[key]
        bn, rows = sv["bn"], sv["rows"]
        co = sv["c"].shape[-1]
        a = self.arena
        gc = self.scratch(slot, sv["c"].shape)
        sums = self.ws.at_least("bn.sums", 2 * co, torch.float32)
        hip.bn_backward_from_stats(self.d, g_masked, sv["c"], stats[0], stats[1], sv["mean"], sv["invstd"], bn.weight, rows,
                                   co, a.grad_flat(bn.weight), a.grad_flat(bn.bias), gc, sums)
        return gc

    def conv_backward(self, key: str, g_c: torch.Tensor, slot: Optional[str], add: Optional[torch.Tensor] = None,
                      add_hw=(0, 0), subgrid: bool = False, fuse_bn: Optional[str] = None,
                      add_bits: Optional[torch.Tensor] = None):
        """Weight gradient into the arena; input gradient (optionally + add) when slot is given.
        fuse_bn=<stage key>: the conv's input was that stage's relu(bn(c)); returns (masked gradient, stats) for
        bn_backward_fused instead of the plain input gradient."""
        sv = self.saved[key]
        geom, conv = sv["geom"], sv["conv"]
        w = conv.weight
        a = self.arena
        if sv["stem_packed"]:
            co = geom["Cout"]
            dwp = self.ws.get(key + ".dwpad", (co, 224), torch.float32)

            def packed_wgrad():
                hip.zero_(dwp)
                work = None
                ring = hip.stemp_wgrad_workspace(self.d, geom["N"], geom["H"], geom["W"], co) if (_CONVP and _DET_WGRAD) else 0
                if ring:
                    # the stem's weight gradient on the LDS ring of image rows (csrc/stemp.hip): one slab per workgroup
                    work = self.ws.at_least("wgrad.slabs." + self._stream_tag(), ring, torch.float32)
                    hip.stemp_wgrad(self.d, g_c, sv["x"], dwp, geom["N"], geom["H"], geom["W"], co, co, work)
                    hip.stem_wfold(self.d, dwp, a.grad_flat(w), co, w.shape[1])
                    return
                if _DET_WGRAD:
                    work = self.ws.at_least("wgrad.slabs." + self._stream_tag(),
                                            hip.stem_wgrad_workspace(self.d, geom["N"], geom["H"], geom["W"], co), torch.float32)
                hip.stem_wgrad(self.d, g_c, sv["x"], dwp, geom["N"], geom["H"], geom["W"], co, co, workspace=work)
                hip.stem_wfold(self.d, dwp, a.grad_flat(w), co, w.shape[1])
            self.on_side(packed_wgrad)
            return None
        if sv["col_input"]:
            kp = geom["Cin"]
            co = geom["Cout"]
            K = w.shape[1] * w.shape[2] * w.shape[3]
            dwp = self.ws.get(key + ".dwpad", (co, kp), torch.float32)

            def stem_wgrad():
                hip.zero_(dwp)
                self.wgrad(g_c, sv["x"], dwp, N=geom["N"], H=1, W=1, Cin=kp, ldx=kp, P=1, Q=1, Cout=co, lddy=co)
                hip.add2d(dwp, a.grad_flat(w), co, K, kp, K)
            self.on_side(stem_wgrad)
            return None
        self.on_side(lambda: self.wgrad(
            g_c, sv["x"], a.grad_flat(w), N=geom["N"], H=geom["H"], W=geom["W"], Cin=geom["Cin"], ldx=geom["ldx"],
            P=geom["P"], Q=geom["Q"], Cout=geom["Cout"], lddy=geom["Cout"], R=geom["R"], S=geom["S"],
            stride=geom["stride"], pad=geom["pad"]))
        if slot is None:
            return None
        N, H, W, ci = geom["N"], geom["H"], geom["W"], geom["Cin"]
        if subgrid:
            # 1x1 stride-2 shortcut: its input gradient is non-zero only on the even (h, w) grid, so it is computed as a
            # plain GEMM on the output grid [N,P,Q] and later folded in by the consumer's epilogue (add_hw)
            dx = self.scratch(slot, (N, geom["P"], geom["Q"], ci))
            hip.conv_gemm(self.d, 0, g_c, self._wd[id(w)], dx, N=N * geom["P"] * geom["Q"], H=1, W=1, Cin=geom["Cout"],
                          ldx=geom["Cout"], P=1, Q=1, Cout=ci, ldy=ci)
            return dx
        dx = self.scratch(slot, (N, H, W, ci))
        if self.s2_classes(conv) and id(w) in self._wd_cls:
            assert add_bits is None      # a stride-2 conv never sits under an identity shortcut
            co, P, Q = geom["Cout"], geom["P"], geom["Q"]
            svp = self.saved[fuse_bn] if fuse_bn is not None else None
            shapes = [((H - (k >> 1) + 1) // 2, (W - (k & 1) + 1) // 2) for k in range(4)]
            stats, tiles_of, total = None, [0] * 4, 0
            if svp is not None:
                assert add is None
                tiles_of = [hip.stat_tiles(self.d, N * pc * qc, ci) if pc > 0 and qc > 0 else 0 for pc, qc in shapes]
                total = sum(tiles_of)
                stats = self.ws.get(fuse_bn + ".bstats", (hip.bn_stats_floats(total, ci),), torch.float32)
            base = 0
            for k in range(4):
                st_k = stats[base * 2 * ci:] if stats is not None else None
                hip.conv_dgrad_s2class(self.d, g_c, self._wd_cls[id(w)][k], dx, add, svp["c"] if svp else None,
                                       svp["scale"] if svp else None, svp["shift"] if svp else None,
                                       svp["mean"] if svp else None, st_k, N, P, Q, co, co, H, W, ci, ci,
                                       ci if add is not None else 0, k >> 1, k & 1, add_hw[0], add_hw[1])
                base += tiles_of[k]
            return (dx, (stats, total)) if svp is not None else dx
        if fuse_bn is not None:
            svp = self.saved[fuse_bn]
            residual = svp.get("bits") is not None       # the fused stage closes a residual block: mask = its bit array
            assert add is None or residual
            if _CONVP and not residual and add is None and self.T == torch.bfloat16 and geom["P"] == H and geom["Q"] == W:
                # interior 3x3 / stride-1 stage: the row-balanced core with the same fused BatchNorm-backward epilogue
                tiles = hip.convp_tiles(self.d, 1, N=N, H=H, W=W, Cin=geom["Cout"], ldx=geom["Cout"], Cout=ci, ldy=ci, R=geom["R"],
                                        S=geom["S"], stride=geom["stride"], pad=geom["pad"])
                if tiles:
                    stats = self.ws.get(fuse_bn + ".bstats", (hip.bn_stats_floats(tiles, ci),), torch.float32)
                    hip.convp_dgrad_bn(self.d, g_c, self._wd[id(w)], dx, svp["c"], svp["scale"], svp["shift"], svp["mean"], stats,
                                       N=N, H=H, W=W, Cin=geom["Cout"], ldx=geom["Cout"], Cout=ci, ldy=ci, tiles=tiles)
                    return dx, (stats, tiles)
            tiles = hip.stat_tiles(self.d, N * H * W, ci)
            stats = self.ws.get(fuse_bn + ".bstats", (hip.bn_stats_floats(tiles, ci),), torch.float32)
            hip.conv_dgrad_bn(self.d, g_c, self._wd[id(w)], dx, svp["c"], svp["scale"], svp["shift"], svp["mean"], stats,
                              N=N, H=geom["P"], W=geom["Q"], Cin=geom["Cout"], ldx=geom["Cout"], P=H, Q=W, Cout=ci, ldy=ci,
                              R=geom["R"], S=geom["S"], stride=geom["stride"], pad=geom["pad"],
                              relu_bits=svp["bits"] if residual else None, add=add,
                              ldadd=ci if add is not None else 0, add_bits=add_bits, add_hw=add_hw)
            return dx, (stats, tiles)
        hip.conv_gemm(self.d, 1, g_c, self._wd[id(w)], dx, N=N, H=geom["P"], W=geom["Q"], Cin=geom["Cout"],
                      ldx=geom["Cout"], P=H, Q=W, Cout=ci, ldy=ci, R=geom["R"], S=geom["S"], stride=geom["stride"],
                      pad=geom["pad"], add=add, ldadd=ci if add is not None else 0, add_hw=add_hw, add_bits=add_bits)
        return dx

    # ------------------------------------------------------------------ transformer ops ----
    def linear(self, key: str, x: torch.Tensor, lin: nn.Linear, train: bool, add: Optional[torch.Tensor] = None,
               row_scale=None) -> torch.Tensor:
        """y = x @ W^T + b (+ add); x: [M, K] in the compute dtype.  row_scale = (per-sample scale [B], rows per sample), with
        add: y = add + scale[m // rows] * (x @ W^T + b) — in the fp8 GEMM's epilogue, a second pass otherwise."""
        M, K = x.shape
        N = lin.weight.shape[0]
        y = self.ws.get(key + ".y", (M, N), self.T)
        bias = self.arena.param_flat(lin.bias) if lin.bias is not None else None
        if self._fp8_linear_ok(lin, M):
            xq, sx, _ = self._fp8_operand(key + ".f8x", x, hip.E4M3)
            wq, _, sw = self._f8w[id(lin.weight)]
            hip.gemm_fp8(0, xq, wq, y, M, K, N, deq_x=sx[1:2], deq_w=sw[1:2], bias=bias, add=add,
                         ldadd=N if add is not None else 0, row_scale=row_scale[0] if row_scale else None,
                         rows_per_sample=row_scale[1] if row_scale else 0)
            if train:
                self.saved[key] = dict(x=x, lin=lin, xq=xq, sx=sx)
            return y
        if row_scale is not None and add is not None and _ROWSCALE_EPILOGUE and self.T == torch.bfloat16 \
                and hip.linear_gelu_fused_ok(self.d, M, K, N):
            # stochastic depth in the residual epilogue of the eight-phase core (no second pass over the branch output)
            hip.linear_residual_scaled(self.d, x, self.w_fwd(lin.weight), bias, add, row_scale[0], row_scale[1], y, M, K, N)
        elif row_scale is not None:
            hip.conv_gemm(self.d, 0, x, self.w_fwd(lin.weight), y, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N, bias=bias)
            out = self.ws.get(key + ".ys", (M, N), self.T)
            hip.scale_rows(self.d, y, add, out, row_scale[0], M // row_scale[1], row_scale[1] * N)
            y = out
        else:
            hip.conv_gemm(self.d, 0, x, self.w_fwd(lin.weight), y, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N,
                          bias=bias, add=add, ldadd=N if add is not None else 0)
        if train:
            self.saved[key] = dict(x=x, lin=lin)
        return y

    @staticmethod
    def _fp8_wgrad_ok(sv, M: int, K: int, N: int) -> bool:
        return _FP8_WGRAD and sv.get("xq") is not None and hip.wgrad_fp8_workspace(M, K, N) > 0

    def _linear_wgrad(self, sv, g: torch.Tensor, gq=None, sg=None, bias_done: bool = False):
        """Weight / bias gradient of a Linear on the side stream.  With both fp8 copies at hand (the forward operand xq and the
        data gradient's operand gq) the contraction runs on the fp8 kernel (NKB_FP8_WGRAD=0: bf16); the bias gradient is then
        the column sum of the unquantised g."""
        x, lin = sv["x"], sv["lin"]
        M, K = x.shape
        N = lin.weight.shape[0]
        a = self.arena
        dbias = a.grad_flat(lin.bias) if lin.bias is not None else None
        if gq is not None and self._fp8_wgrad_ok(sv, M, K, N):
            xq, sx = sv["xq"], sv["sx"]

            def run():
                work = self.ws.at_least("wgrad.slabs." + self._stream_tag(), hip.wgrad_fp8_workspace(M, K, N), torch.float32)
                hip.wgrad_fp8(gq, xq, a.grad_flat(lin.weight), M, K, N, deq_g=sg[1:2], deq_x=sx[1:2], workspace=work)
                if dbias is not None and not bias_done:
                    self.colsum2d(g, dbias, M, N, N)
            self.on_side(run)
            return
        self.on_side(lambda: self.wgrad(g, x, a.grad_flat(lin.weight), N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N,
                                        dbias=dbias))

    def _branch_gradient(self, sv, lin, g: torch.Tensor, g_scale, M: int, K: int, N: int, slot: str):
        """g_scale = (per-sample scale, rows per sample) of a stochastic-depth branch: the gradient that enters the Linear is
        scale[m // rows] * g.  Returns (g, row_scale): either the scaled tensor materialised in bf16 (row_scale None), or g itself
        with the scale handed on to the fused fp8 quantise + column-sum pass (no bf16 copy of the branch gradient at all)."""
        if g_scale is None:
            return g, None
        ready = self._gs_ready.pop(g.data_ptr(), None)        # LayerNorm backward wrote the scaled copy next to g
        if ready is not None:
            return ready, None
        if (self._fp8_linear_ok(lin, M) and lin.bias is not None and self._fp8_wgrad_ok(sv, M, K, N) and self._fp8_colsum_ok(g)
                and _FP8_FUSED_QUANT):
            return g, g_scale
        dx = self.scratch(slot, g.shape)
        hip.scale_rows(self.d, g, None, dx, g_scale[0], M // g_scale[1], g_scale[1] * g.shape[1])
        return dx, None

    def linear_backward(self, key: str, g: torch.Tensor, slot: Optional[str], add: Optional[torch.Tensor] = None, g_scale=None):
        sv = self.saved[key]
        x, lin = sv["x"], sv["lin"]
        M, K = x.shape
        N = lin.weight.shape[0]
        fp8 = self._fp8_linear_ok(lin, M)
        g, rsc = self._branch_gradient(sv, lin, g, g_scale, M, K, N, "gs_" + key.rsplit(".", 1)[-1])   # (read later by the side stream)
        gq = sg = None
        bias_done = False
        if fp8 and (slot is not None or (_FP8_WGRAD and sv.get("xq") is not None)):
            # one e5m2 copy serves the data and the weight gradient; the pass that makes it also sums the columns (bias gradient)
            want = self.arena.grad_flat(lin.bias) if (lin.bias is not None and self._fp8_wgrad_ok(sv, M, K, N)) else None
            gq, sg, bias_done = self._fp8_operand(key + ".f8g", g, hip.E5M2, colsum=want, row_scale=rsc)
        self._linear_wgrad(sv, g, gq, sg, bias_done)
        if slot is None:
            return None
        dx = self.scratch(slot, (M, K))
        if fp8:
            _, wdq, sw = self._f8w[id(lin.weight)]
            hip.gemm_fp8(1, gq, wdq, dx, M, N, K, deq_x=sg[1:2], deq_w=sw[1:2], add=add, ldadd=K if add is not None else 0)
        else:
            hip.conv_gemm(self.d, 0, g, self._wd[id(lin.weight)], dx, N=M, H=1, W=1, Cin=N, ldx=N, P=1, Q=1, Cout=K, ldy=K,
                          add=add, ldadd=K if add is not None else 0)
        return dx

    def linear_relu6(self, key: str, x: torch.Tensor, lin: nn.Linear, train: bool, q_for: Optional[str] = None,
                     consumer: Optional[nn.Linear] = None) -> torch.Tensor:
        """u = relu6(x @ W^T + b) with the clamp in the GEMM epilogue (unicom Mlp: fc1 -> ReLU6); only u is kept.
        q_for: the fp8 site (key of the Linear that consumes u + ".f8x") whose operand the epilogue writes as well.
        consumer: that Linear.  When its forward, data gradient AND weight gradient all run on the fp8 copy, nothing reads u in
        bf16 any more except the ReLU6 mask of the backward pass: the epilogue then writes the mask as bits (1/16 of the bytes)
        and does not store u at all — the returned tensor is a placeholder that only carries the shape."""
        M, K = x.shape
        N = lin.weight.shape[0]
        u = self.ws.get(key + ".y", (M, N), self.T)
        bias = self.arena.param_flat(lin.bias) if lin.bias is not None else None
        if self._fp8_linear_ok(lin, M):
            xq, sx, _ = self._fp8_operand(key + ".f8x", x, hip.E4M3)
            wq, _, sw = self._f8w[id(lin.weight)]
            out = self._fp8_produce(q_for, (M, N), hip.E4M3) if train else None
            bits = None
            if (out and _FP8_MASK_BITS and consumer is not None and _FP8_WGRAD and M % 128 == 0 and N % 8 == 0
                    and self._fp8_linear_ok(consumer, M) and hip.wgrad_fp8_workspace(M, N, consumer.weight.shape[0]) > 0):
                bits = self.ws.get(key + ".bits", (M, N // 8), torch.uint8)
            hip.gemm_fp8(0, xq, wq, None if bits is not None else u, M, K, N, deq_x=sx[1:2], deq_w=sw[1:2], bias=bias, relu=2,
                         yq=out[0] if out else None, q_state=out[1] if out else None, q_kind=out[2] if out else 0, mask_out=bits)
            if train:
                self.saved[key] = dict(x=x, lin=lin, u=None if bits is not None else u, ubits=bits, xq=xq, sx=sx)
            return u
        else:
            hip.conv_gemm(self.d, 0, x, self.w_fwd(lin.weight), u, N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, ldy=N,
                          bias=bias, relu=2)
        if train:
            self.saved[key] = dict(x=x, lin=lin, u=u)
        return u

    def linear_backward_through_relu6(self, key_next: str, key_act: str, g: torch.Tensor, slot: str,
                                      q_for: Optional[str] = None, g_scale=None) -> torch.Tensor:
        """For u = relu6(pre), y = u @ W2^T + b2: weight/bias gradient of W2 (side stream) and d_pre = (g @ W2) masked by
        0 < u < 6 in one GEMM epilogue."""
        sv = self.saved[key_next]
        x, lin = sv["x"], sv["lin"]          # x = u (ReLU6 output), lin = fc2
        M, K = x.shape
        N = lin.weight.shape[0]
        fp8 = self._fp8_linear_ok(lin, M)
        g, rsc = self._branch_gradient(sv, lin, g, g_scale, M, K, N, "gs_" + key_next.rsplit(".", 1)[-1])
        gq = sg = None
        bias_done = False
        if fp8:
            want = self.arena.grad_flat(lin.bias) if (lin.bias is not None and self._fp8_wgrad_ok(sv, M, K, N)) else None
            gq, sg, bias_done = self._fp8_operand(key_next + ".f8g", g, hip.E5M2, colsum=want, row_scale=rsc)
        self._linear_wgrad(sv, g, gq, sg, bias_done)
        d_pre = self.scratch(slot, (M, K))
        if fp8:
            _, wdq, sw = self._f8w[id(lin.weight)]
            out = self._fp8_produce(q_for, (M, K), hip.E5M2)
            sva = self.saved[key_act]
            ubits = sva.get("ubits") if out else None         # (the bit form needs the quantised second output)
            if ubits is None and sva["u"] is None:
                raise RuntimeError("the ReLU6 output was kept as mask bits only, but this data gradient cannot consume them")
            # When the Linear before the ReLU6 (fc1) takes its data AND weight gradient from the fp8 copy, the only other reader of
            # d_pre is its bias gradient: the epilogue sums the columns too and d_pre is never stored in bf16.
            lin1 = sva["lin"]
            csum = work = None
            if (_FP8_EPI_COLSUM and ubits is not None and lin1.bias is not None and M % 256 == 0
                    and self._fp8_wgrad_ok(sva, M, lin1.weight.shape[1], K)):
                csum = self.arena.grad_flat(lin1.bias)
                work = self.ws.at_least("f8.epicolsum", (M // 256) * K, torch.float32)
                self._f8bias[q_for] = True
            hip.gemm_fp8(1, gq, wdq, None if csum is not None else d_pre, M, N, K, deq_x=sg[1:2], deq_w=sw[1:2],
                         aux=None if ubits is not None else sva["u"], aux_mode=1, mask_in=ubits, colsum=csum, colsum_work=work,
                         yq=out[0] if out else None, q_state=out[1] if out else None, q_kind=out[2] if out else 0)
        else:
            hip.linear_gelu(self.d, 3, g, self._wd[id(lin.weight)], None, self.saved[key_act]["u"], d_pre, None, M, N, K)
        return d_pre

    def linear_gelu(self, key: str, x: torch.Tensor, lin: nn.Linear, train: bool) -> torch.Tensor:
        """u = gelu(x @ W^T + b) with the GELU in the GEMM epilogue; the pre-activation is kept for backward."""
        M, K = x.shape
        N = lin.weight.shape[0]
        pre = self.ws.get(key + ".pre", (M, N), self.T)
        u = self.ws.get(key + ".y", (M, N), self.T)
        hip.linear_gelu(self.d, 1, x, self.w_fwd(lin.weight), self.arena.param_flat(lin.bias), None, u, pre, M, K, N)
        if train:
            self.saved[key] = dict(x=x, lin=lin, pre=pre)
        return u

    def linear_gelu_keep_derivative(self, key: str, key_act: str, x: torch.Tensor, lin: nn.Linear, train: bool):
        """u = gelu(x @ W^T + b) AND gelu'(pre) from the fc1 GEMM's epilogue (the pre-activation is never stored; the separate
        elementwise pass over it is gone); saved like linear() + gelu(keep_derivative=True), so the backward pass is
        linear_backward_through_saved_derivative + linear_backward.  Returns None when the shape is not one of the eight-phase
        core's (the caller then takes the two-kernel path)."""
        M, K = x.shape
        N = lin.weight.shape[0]
        if not (_GELU_EPILOGUE and train and self.T == torch.bfloat16 and lin.bias is not None and not self._fp8_linear_ok(lin, M)
                and hip.linear_gelu_fused_ok(self.d, M, K, N)):
            return None
        u = self.ws.get(key_act + ".y", (M, N), self.T)
        gp = self.ws.get(key + ".y", (M, N), self.T)            # (the buffer the plain path keeps the pre-activation / derivative in)
        hip.linear_gelu(self.d, 5, x, self.w_fwd(lin.weight), self.arena.param_flat(lin.bias), None, u, gp, M, K, N)
        self.saved[key] = dict(x=x, lin=lin)
        self.saved[key_act] = dict(gp=gp)
        return u

    def linear_backward_through_gelu(self, key_next: str, key_act: str, g: torch.Tensor, slot: str) -> torch.Tensor:
        """For u = gelu(pre), y = u @ W2^T + b2: weight/bias gradient of W2 (side stream) and d_pre = (g @ W2) * gelu'(pre)
        in one GEMM epilogue (the separate GELU-backward pass and the d_u tensor disappear)."""
        sv = self.saved[key_next]
        x, lin = sv["x"], sv["lin"]          # x = u (gelu output), lin = fc2
        M, K = x.shape                      # K = hidden width
        N = lin.weight.shape[0]
        a = self.arena
        self.on_side(lambda: self.wgrad(
            g, x, a.grad_flat(lin.weight), N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N,
            dbias=a.grad_flat(lin.bias) if lin.bias is not None else None))
        d_pre = self.scratch(slot, (M, K))
        hip.linear_gelu(self.d, 2, g, self._wd[id(lin.weight)], None, self.saved[key_act]["pre"], d_pre, None, M, N, K)
        return d_pre

    def layernorm(self, key: str, x: torch.Tensor, ln: nn.LayerNorm, train: bool, rows: Optional[int] = None,
                  x_stride: Optional[int] = None, q_for: Optional[str] = None) -> torch.Tensor:
        D = ln.weight.shape[0]
        rows = x.shape[0] if rows is None else rows
        xs = D if x_stride is None else x_stride
        y = self.ws.get(key + ".y", (rows, D), self.T)
        st = self.ws.get(key + ".stat", (2, rows), torch.float32)
        a = self.arena
        # q_for: the fp8 site (consuming Linear's key + ".f8x") whose operand this kernel writes next to y
        out = self._fp8_produce(q_for, (rows, D), hip.E4M3) if (train and self.fp8 and self.T == torch.bfloat16 and D % 256 == 0) else None
        hip.layernorm_fwd(self.d, x, xs, a.param_flat(ln.weight), a.param_flat(ln.bias), y, D, st[0], st[1], rows, D, ln.eps,
                          yq=out[0] if out else None, q_state=out[1] if out else None, q_kind=out[2] if out else 0)
        if train:
            self.saved[key] = dict(x=x, xs=xs, rows=rows, ln=ln, mean=st[0], rstd=st[1])
        return y

    def layernorm_backward(self, key: str, g: torch.Tensor, out: torch.Tensor, out_stride: int,
                           add: Optional[torch.Tensor] = None, consumer: Optional[str] = None, consumer_dp: Optional[str] = None,
                           consumer_block: Optional[int] = None):
        """dx (+ add) is written into `out` rows with stride out_stride (elements).
        consumer: key of the Linear whose backward pass takes `out` as its incoming gradient (through the stochastic-depth site
        consumer_dp, if any).  In the fp8 step that Linear consumes the gradient as an e5m2 operand scaled per sample, plus its
        column sums for the bias: this kernel then writes both next to dx (no separate quantise + column-sum pass over dx).
        In the bf16 step it writes the scaled branch gradient scale[b] * dx itself (consumer_block: the backward block index the
        consumer runs in, when that is not the current one — its scratch set and the side-stream wait that guards it)."""
        sv = self.saved[key]
        ln = sv["ln"]
        D = ln.weight.shape[0]
        rows = sv["rows"]
        a = self.arena
        # The parameter-gradient half (two small launches that sum the per-block partial rows) only feeds dgamma / dbeta (/ a bias
        # gradient): it goes to the side stream with the weight gradients instead of sitting in the backward chain.  Its partial
        # rows then live in a workspace of this LayerNorm's own (per block parity, like the gradient scratch: begin_block waits
        # for the side-stream readers before a set is reused).
        split = _LN_REDUCE_SIDE and self.overlap_wgrad
        work = self.ws.at_least("ln.work" + (self._suffix + "." + key.rsplit(".", 1)[-1] if split else ""), hip.layernorm_ws(D),
                                torch.float32)
        dg, db = a.grad_flat(ln.weight), a.grad_flat(ln.bias)

        def launch(colsum=None, **kw):
            if not split:
                hip.layernorm_bwd(self.d, g, D, sv["x"], sv["xs"], a.param_flat(ln.weight), sv["mean"], sv["rstd"], add, out,
                                  out_stride, dg, db, rows, D, workspace=work, colsum=colsum, **kw)
                return
            hip.layernorm_bwd(self.d, g, D, sv["x"], sv["xs"], a.param_flat(ln.weight), sv["mean"], sv["rstd"], add, out,
                              out_stride, None, None, rows, D, workspace=work, **kw)
            self.on_side(lambda: hip.layernorm_param_reduce(work, rows, D, 3 if colsum is not None else 2, dg, db, colsum))
        q = None
        if consumer is not None and self.fp8 and _FP8_LN_BWD_QUANT and _FP8_FUSED_QUANT and self.T == torch.bfloat16:
            svc = self.saved.get(consumer)
            if svc is not None and out_stride == D and D % 256 == 0 and tuple(out.shape) == (rows, D):
                lin = svc["lin"]
                Mc, Kc = svc["x"].shape
                Nc = lin.weight.shape[0]
                if (Mc == rows and Nc == D and lin.bias is not None and self._fp8_linear_ok(lin, Mc)
                        and self._fp8_wgrad_ok(svc, Mc, Kc, Nc) and self._fp8_colsum_ok(out)):
                    q = self._fp8_produce(consumer + ".f8g", (rows, D), hip.E5M2)
                    if q is not None:
                        gs = self.drop_path_gscale(consumer_dp, rows) if consumer_dp else None
                        self._f8bias[consumer + ".f8g"] = True
                        launch(colsum=a.grad_flat(lin.bias), yq=q[0], q_state=q[1], q_kind=q[2], row_scale=gs[0] if gs else None,
                               rows_per_sample=gs[1] if gs else 0)
                        return out
        if (consumer is not None and consumer_dp and _LN_BWD_SCALED_COPY and self.T == torch.bfloat16 and out_stride == D
                and D % 256 == 0 and tuple(out.shape) == (rows, D) and consumer in self.saved):
            gs = self.drop_path_gscale(consumer_dp, rows)
            svc = self.saved[consumer]
            if gs is not None and not self._fp8_linear_ok(svc["lin"], rows) and svc["lin"].weight.shape[0] == D:
                # bf16 step: the branch gradient scale[b] * dx that the consumer's data and weight gradient read is written here too
                # (it was a scale_rows pass over dx: read + write of the whole tensor)
                keep = self._suffix
                if consumer_block is not None:
                    # the consumer's scratch set (by block parity) is still read by the side-stream weight gradients issued two
                    # blocks before it: the wait begin_block(consumer_block) would make, made now
                    ev = self._side_done.pop(consumer_block + 2, None)
                    if ev is not None:
                        hip.stream_wait_event(torch.cuda.current_stream(), ev)
                    self._suffix = ".p%d" % (consumer_block & 1)
                scaled = self.scratch("gs_" + consumer.rsplit(".", 1)[-1], (rows, D))
                self._suffix = keep
                launch(yq=scaled, q_kind=2, row_scale=gs[0], rows_per_sample=gs[1])
                self._gs_ready[out.data_ptr()] = scaled
                return out
        launch()
        return out

    def gelu(self, key: str, x: torch.Tensor, train: bool, keep_derivative: bool = False) -> torch.Tensor:
        """keep_derivative: the forward pass also stores gelu'(x), overwriting x (the pre-activation is not needed again);
        the backward pass is then linear_backward_through_saved_derivative instead of gelu_backward."""
        y = self.ws.get(key + ".y", x.shape, self.T)
        if train and keep_derivative:
            hip.gelu_fwd_dgelu(self.d, x, y, x, x.numel())
            self.saved[key] = dict(gp=x)
            return y
        hip.gelu(self.d, x, None, y, x.numel())
        if train:
            self.saved[key] = dict(x=x)
        return y

    def linear_backward_through_saved_derivative(self, key_next: str, key_act: str, g: torch.Tensor, slot: str) -> torch.Tensor:
        """For u = act(pre), y = u @ W2^T + b2 with act'(pre) kept by the forward pass: weight/bias gradient of W2 (side
        stream) and d_pre = (g @ W2) * act'(pre) in one GEMM epilogue."""
        sv = self.saved[key_next]
        x, lin = sv["x"], sv["lin"]
        M, K = x.shape
        N = lin.weight.shape[0]
        a = self.arena
        self.on_side(lambda: self.wgrad(
            g, x, a.grad_flat(lin.weight), N=M, H=1, W=1, Cin=K, ldx=K, P=1, Q=1, Cout=N, lddy=N,
            dbias=a.grad_flat(lin.bias) if lin.bias is not None else None))
        d_pre = self.scratch(slot, (M, K))
        hip.linear_gelu(self.d, 4, g, self._wd[id(lin.weight)], None, self.saved[key_act]["gp"], d_pre, None, M, N, K)
        return d_pre

    def gelu_backward(self, key: str, g: torch.Tensor, slot: str) -> torch.Tensor:
        x = self.saved[key]["x"]
        dx = self.scratch(slot, x.shape)
        hip.gelu(self.d, x, g, dx, x.numel())
        return dx

    def relu6(self, key: str, x: torch.Tensor, train: bool) -> torch.Tensor:
        y = self.ws.get(key + ".y", x.shape, self.T)
        hip.relu6(self.d, x, None, y, x.numel())
        if train:
            self.saved[key] = dict(x=x)
        return y

    def relu6_backward(self, key: str, g: torch.Tensor, slot: str) -> torch.Tensor:
        x = self.saved[key]["x"]
        dx = self.scratch(slot, x.shape)
        hip.relu6(self.d, x, g, dx, x.numel())
        return dx

    def _ones(self, n: int) -> torch.Tensor:
        """A persistent vector of ones (never written after its creation: one fill per engine, not one per drop-path site)."""
        t = self._ones_cache.get(n)
        if t is None:
            t = self._ones_cache[n] = torch.ones(n, device=self.device, dtype=torch.float32)
        return t

    def drop_path_scale(self, key: str, p: float, samples: int) -> torch.Tensor:
        """The per-sample factor keep[b] / (1 - p) of one stochastic-depth site (kept for the backward pass)."""
        ones = self._ones(samples)
        scale = self.ws.get(key + ".scale", (samples,), torch.float32)
        mask = self.ws.get(key + ".mask", (samples,), torch.uint8)
        seed = hip.fresh_seed()
        hip.dropout(hip.F32, False, ones, None, scale, mask, samples, p, seed)       # scale[b] = keep / (1 - p)
        self.saved[key] = dict(scale=scale, samples=samples)
        return scale

    def drop_path_gscale(self, key: str, rows: int):
        """(scale, rows per sample) of the stochastic-depth site `key` for the backward pass; None when it was inactive."""
        sv = self.saved.get(key)
        return None if sv is None else (sv["scale"], rows // sv["samples"])

    def drop_path(self, key: str, x: torch.Tensor, p: float, train: bool, samples: int, add: torch.Tensor) -> torch.Tensor:
        """Stochastic depth on a residual branch: y = add + x * keep[b] / (1 - p), one Bernoulli(1 - p) draw per sample
        (timm-style DropPath as used by the unicom blocks).  Only called when active (train and p > 0)."""
        assert train and p > 0
        ones = self._ones(samples)
        scale = self.ws.get(key + ".scale", (samples,), torch.float32)
        mask = self.ws.get(key + ".mask", (samples,), torch.uint8)
        seed = hip.fresh_seed()
        hip.dropout(hip.F32, False, ones, None, scale, mask, samples, p, seed)       # scale[b] = keep / (1 - p)
        y = self.ws.get(key + ".y", x.shape, self.T)
        hip.scale_rows(self.d, x, add, y, scale, samples, x.numel() // samples)
        self.saved[key] = dict(scale=scale, samples=samples)
        return y

    def drop_path_backward(self, key: str, g: torch.Tensor, slot: str) -> torch.Tensor:
        """Branch gradient through drop_path(key); a pass-through when it was inactive in the forward pass."""
        sv = self.saved.get(key)
        if sv is None:
            return g
        dx = self.scratch(slot, g.shape)
        hip.scale_rows(self.d, g, None, dx, sv["scale"], sv["samples"], g.numel() // sv["samples"])
        return dx

    def dropout(self, key: str, x: torch.Tensor, p: float, train: bool, add: Optional[torch.Tensor] = None) -> torch.Tensor:
        """nn.Dropout(p) in train mode (+ add): y = keep * x / (1 - p) (+ add).  Identity (+ add) otherwise.  The keep mask
        (one byte per element, counter-based generator seeded from torch's global generator) is kept for backward."""
        if not (train and p > 0):
            if add is None:
                return x
            y = self.ws.get(key + ".y", x.shape, self.T)
            hip.host_op(lambda: torch.add(x, add, out=y))
            return y
        y = self.ws.get(key + ".y", x.shape, x.dtype)
        mask = self.ws.get(key + ".mask", x.shape, torch.uint8)
        seed = hip.fresh_seed()
        hip.dropout(hip.dt(x.dtype), False, x, add, y, mask, x.numel(), p, seed)
        self.saved[key] = dict(mask=mask, p=p)
        return y

    def dropout_backward(self, key: str, g: torch.Tensor, slot: str) -> torch.Tensor:
        """Gradient through dropout(key); a pass-through when that dropout was inactive in the forward pass."""
        sv = self.saved.get(key)
        if sv is None:
            return g
        dx = self.ws.get("grad.%s%s.%s" % (slot, self._suffix, "x".join(str(int(v)) for v in g.shape)), g.shape, g.dtype)
        hip.dropout(hip.dt(g.dtype), True, g, None, dx, sv["mask"], g.numel(), sv["p"], 0)
        return dx

    def attention(self, key: str, qkv: torch.Tensor, B: int, T: int, H: int, train: bool, drop_p: float = 0.0,
                  q_for: Optional[str] = None) -> torch.Tensor:
        """softmax(q k^T / sqrt(dh)) v over all (image, head) pairs; qkv: [B*T, 3*D] laid out [which][head][dh].
        drop_p: attention-probability dropout (timm Attention.attn_drop); active dropout takes the materialised path."""
        D = qkv.shape[1] // 3
        dh = D // H
        Tp = _round_up(T, self.kte)
        attn_drop = train and drop_p > 0
        if self.fused_attention and self.T == torch.bfloat16 and dh == 64 and T <= 256 and not attn_drop:
            # fused kernel: scores and probabilities never reach HBM; only the per-row log-sum-exp is kept
            o = self.ws.get(key + ".o", (B * T, D), self.T)
            lse = self.ws.get(key + ".lse", (B * H, T), torch.float32)
            out = self._fp8_produce(q_for, (B * T, D), hip.E4M3) if (train and self.fp8) else None    # the projection's fp8 operand
            hip.attn_forward(self.d, qkv, o, lse, B, T, H, dh, dh ** -0.5, outq=out[0] if out else None,
                             q_state=out[1] if out else None)
            if train:
                self.saved[key] = dict(qkv=qkv, lse=lse, o=o, B=B, T=T, H=H, fused=True)
            return o
        S = self.ws.get("attn.S", (B * H, T, Tp), torch.float32)
        P = self.ws.get(key + ".P", (B * H, T, Tp), self.T)
        Vt = self.ws.get("attn.Vt", (B * H, dh, Tp), self.T)
        o = self.ws.get(key + ".o", (B * T, D), self.T)
        q, k, v = qkv, qkv[:, D:], qkv[:, 2 * D:]
        sq = (T * 3 * D, dh)
        hip.gemm_batched(self.d, q, k, S, T, T, dh, 3 * D, 3 * D, Tp, B, H, sq, sq, (H * T * Tp, T * Tp), out_f32=True)
        hip.attn_softmax(self.d, False, S, Tp, None, P, Tp, B * H * T, T, dh ** -0.5)
        Pd = self.dropout(key + ".drop", P, drop_p, train) if attn_drop else P
        hip.head_transpose(self.d, v, 3 * D, T * 3 * D, dh, B, H, Vt, T, dh, Tp)
        hip.gemm_batched(self.d, Pd, Vt, o, T, dh, Tp, Tp, Tp, D, B, H, (H * T * Tp, T * Tp), (H * dh * Tp, dh * Tp),
                         (T * D, dh))
        if train:
            self.saved[key] = dict(qkv=qkv, P=P, Pd=Pd, B=B, T=T, H=H, drop=attn_drop)
        return o

    def attention_backward(self, key: str, d_o: torch.Tensor, slot: str, q_for: Optional[str] = None) -> torch.Tensor:
        sv = self.saved[key]
        qkv, B, T, H = sv["qkv"], sv["B"], sv["T"], sv["H"]
        D = qkv.shape[1] // 3
        dh = D // H
        Tp = _round_up(T, self.kte)
        dqkv = self.scratch(slot, qkv.shape)
        Kt = self.ws.get("attn.Vt", (B * H, dh, Tp), self.T)
        q, k, v = qkv, qkv[:, D:], qkv[:, 2 * D:]
        sq, sp, so = (T * 3 * D, dh), (H * T * Tp, T * Tp), (T * D, dh)
        if sv.get("fused") and _ATTN_FUSED_BWD:
            # dQ, dK, dV in one kernel per layer: P and dS never reach HBM
            out = self._fp8_produce(q_for, tuple(qkv.shape), hip.E5M2) if self.fp8 else None       # the qkv gradients' fp8 operand
            csum = work = None
            if out and _FP8_ATTN_COLSUM and q_for.endswith(".f8g") and q_for[:-4] in self.saved:
                # the qkv projection takes both gradients from the fp8 copy: its bias gradient — the column sums of d_qkv — is the
                # only other reader of the bf16 tensor, and comes out of this kernel's stores instead of a pass over d_qkv
                svc = self.saved[q_for[:-4]]
                lin = svc["lin"]
                Mc, Kc = svc["x"].shape
                if (lin.bias is not None and Mc == B * T and lin.weight.shape[0] == 3 * D and self._fp8_linear_ok(lin, Mc)
                        and self._fp8_wgrad_ok(svc, Mc, Kc, 3 * D)):
                    csum = self.arena.grad_flat(lin.bias)
                    work = self.ws.at_least("attn.colsum", B * 3 * D, torch.float32)
                    self._f8bias[q_for] = True
            hip.attn_backward(self.d, qkv, d_o, sv["o"], sv["lse"], dqkv, B, T, H, dh, dh ** -0.5,
                              dqkv_q=out[0] if out else None, q_state=out[1] if out else None, colsum=csum, colsum_work=work)
            return dqkv
        if sv.get("fused"):
            # P and dS are recomputed in one pass (pad columns beyond roundup(T,16) stay zero from allocation)
            P = self.ws.get("attn.Pbwd", (B * H, T, Tp), self.T, zero=True)
            dS = self.ws.get("attn.dS", (B * H, T, Tp), self.T, zero=True)
            # the same pass forms dQ = dS K (into the Q third of dqkv); dV and dK stay batched GEMMs on P / dS
            hip.attn_backward_ds(self.d, qkv, d_o, sv["lse"], P, dS, Tp, B, T, H, dh, dh ** -0.5,
                                 dq=dqkv if _ATTN_FUSED_DQ else None, ld_dq=3 * D)
            hip.gemm_tn_batched(self.d, P, d_o, dqkv[:, 2 * D:], T, T, dh, Tp, D, 3 * D, B, H, sp, so, sq)
            if _ATTN_FUSED_DQ:
                hip.gemm_tn_batched(self.d, dS, q, dqkv[:, D:], T, T, dh, Tp, 3 * D, 3 * D, B, H, sp, sq, sq)
                return dqkv
        else:
            P = sv["P"]
            dP = self.ws.get("attn.S", (B * H, T, Tp), torch.float32)
            dS = self.ws.get("attn.dS", (B * H, T, Tp), self.T)
            # dV = Pd^T dO   (Pd = dropped probabilities when attention dropout was active, else P itself)
            hip.gemm_tn_batched(self.d, sv["Pd"], d_o, dqkv[:, 2 * D:], T, T, dh, Tp, D, 3 * D, B, H, sp, so, sq)
            # dPd = dO V^T ; dP = dropout'(dPd) ; dS = softmax'(P, dP)
            hip.gemm_batched(self.d, d_o, v, dP, T, T, dh, D, 3 * D, Tp, B, H, so, sq, sp, out_f32=True)
            if sv["drop"]:
                dsv = self.saved[key + ".drop"]
                hip.dropout(hip.dt(torch.float32), True, dP, None, dP, dsv["mask"], dP.numel(), dsv["p"], 0)
            hip.attn_softmax(self.d, True, dP, Tp, P, dS, Tp, B * H * T, T, dh ** -0.5)
        # dQ = dS K ; dK = dS^T Q
        hip.head_transpose(self.d, k, 3 * D, T * 3 * D, dh, B, H, Kt, T, dh, Tp)
        hip.gemm_batched(self.d, dS, Kt, dqkv, T, dh, Tp, Tp, Tp, 3 * D, B, H, sp, (H * dh * Tp, dh * Tp), sq)
        hip.gemm_tn_batched(self.d, dS, q, dqkv[:, D:], T, T, dh, Tp, 3 * D, 3 * D, B, H, sp, sq, sq)
        return dqkv
