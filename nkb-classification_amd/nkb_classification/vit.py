"""timm VisionTransformer family (vit_base_patch16_224 layout, num_classes=0, class-token pooling) for the HIP
engine: parameter containers with timm's state-dict names + the forward / backward execution plan.

Reference call site: timm.create_model("vit_base_patch16_224", ...) at
/root/reference/nkb_classification/model.py:82; architecture per SURVEY.md §8 A8 (LayerNorm eps 1e-6, qkv with
bias, exact-erf GELU, pre-norm residual blocks, x[:, 0] of the final norm as the embedding).
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import hip
from .backbones import _ParamOnly
from .hipnet import HipEngine

_GELU_KEEP_DERIV = True   # forward stores gelu'(pre); backward = fc2-dgrad epilogue multiply
_FUSED_GELU = False   # measured: erf in the GEMM epilogue costs more than the pass it saves (66.3 vs 65.9 ms)


class _PatchEmbed(_ParamOnly):
    def __init__(self, patch, in_chans, dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, patch, patch)


class _Attention(_ParamOnly):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)


class _Mlp(_ParamOnly):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.drop1 = nn.Dropout(0.0)
        self.fc2 = nn.Linear(hidden, dim)
        self.drop2 = nn.Dropout(0.0)


class _Block(_ParamOnly):
    def __init__(self, dim, heads, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class HipViT(_ParamOnly):
    family = "vit"

    def __init__(self, img=224, patch=16, dim=768, depth=12, heads=12, mlp_ratio=4.0):
        super().__init__()
        self.num_features = dim
        self.img, self.patch, self.heads = img, patch, heads
        self.patch_embed = _PatchEmbed(patch, 3, dim)
        n_tok = (img // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n_tok + 1, dim) * 0.02)
        self.pos_drop = nn.Dropout(0.0)
        self.blocks = nn.Sequential(*[_Block(dim, heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head_drop = nn.Dropout(0.0)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def gemm_convs(self):
        return [m for m in self.modules() if isinstance(m, nn.Linear)]

    def stem_convs(self):
        return [self.patch_embed.proj]

    def fp8_linears(self):
        """The Linear layers of the transformer blocks: the contractions that cfg.amp_dtype = "fp8" moves to fp8 operands."""
        return [m for blk in self.blocks for m in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2)]

    def run_forward(self, eng: HipEngine, img: torch.Tensor, train: bool) -> torch.Tensor:
        """Dropout sites follow timm's VisionTransformer (every nn.Dropout the reference's set_dropout rewrites,
        model.py:66-72): pos_drop after the position embedding, attn_drop on the attention probabilities, proj_drop and
        mlp.drop2 before the residual additions, mlp.drop1 after the GELU, head_drop on the pooled embedding."""
        for k in [k for k in eng.saved if k.endswith(".drop") or k.endswith("_drop")]:
            del eng.saved[k]                                 # masks of a previous step must not leak into this backward
        B, C, Hh, Ww = img.shape
        if Hh != self.img or Ww != self.img:
            raise RuntimeError(f"this ViT expects {self.img}x{self.img} inputs (pos_embed is fixed), got {Hh}x{Ww}")
        pr = self.patch_embed.proj
        D, ps = self.num_features, self.patch
        gh = Hh // ps
        npatch = gh * gh
        T = npatch + 1
        K = C * ps * ps
        kp = eng.kpad(K)
        col = eng.ws.get("pe.col", (B * npatch, kp), eng.T)
        hip.im2row(eng.d, img, col, B, C, Hh, Ww, ps, ps, ps, 0, kp)
        tok = eng.ws.get("pe.tok", (B * npatch, D), eng.T)
        a = eng.arena
        hip.conv_gemm(eng.d, 0, col, eng.w_fwd(pr.weight), tok, N=B * npatch, H=1, W=1, Cin=kp, ldx=kp, P=1, Q=1, Cout=D,
                      ldy=D, bias=a.param_flat(pr.bias))
        x = eng.ws.get("pe.x", (B * T, D), eng.T)
        hip.vit_assemble(eng.d, False, tok, a.param_flat(self.cls_token), a.param_flat(self.pos_embed), x, B, T, D)
        if train:
            eng.saved["pe"] = dict(col=col, B=B, T=T, kp=kp, K=K)
        x = eng.dropout("pos_drop", x, self.pos_drop.p, train)
        for i, blk in enumerate(self.blocks):
            at, mlp = blk.attn, blk.mlp
            h = eng.layernorm(f"b{i}.ln1", x, blk.norm1, train, q_for=f"b{i}.qkv.f8x")
            qkv = eng.linear(f"b{i}.qkv", h, at.qkv, train)
            o = eng.attention(f"b{i}.attn", qkv, B, T, self.heads, train, drop_p=at.attn_drop.p, q_for=f"b{i}.proj.f8x")
            if train and at.proj_drop.p > 0:
                x = eng.dropout(f"b{i}.proj_drop", eng.linear(f"b{i}.proj", o, at.proj, train), at.proj_drop.p, train, add=x)
            else:
                x = eng.linear(f"b{i}.proj", o, at.proj, train, add=x)
            h = eng.layernorm(f"b{i}.ln2", x, blk.norm2, train, q_for=f"b{i}.fc1.f8x")
            if _FUSED_GELU:
                u = eng.linear_gelu(f"b{i}.fc1", h, mlp.fc1, train)        # GELU fused into the fc1 epilogue
            else:
                keep = _GELU_KEEP_DERIV and not (train and mlp.drop1.p > 0)
                u = eng.linear_gelu_keep_derivative(f"b{i}.fc1", f"b{i}.act", h, mlp.fc1, train) if keep else None
                if u is None:
                    u = eng.gelu(f"b{i}.act", eng.linear(f"b{i}.fc1", h, mlp.fc1, train), train, keep_derivative=keep)
            u = eng.dropout(f"b{i}.mlp_drop", u, mlp.drop1.p, train)
            if train and mlp.drop2.p > 0:
                x = eng.dropout(f"b{i}.mlp2_drop", eng.linear(f"b{i}.fc2", u, mlp.fc2, train), mlp.drop2.p, train, add=x)
            else:
                x = eng.linear(f"b{i}.fc2", u, mlp.fc2, train, add=x)
        # final norm on the class-token rows only (x[:, 0]); the other rows never reach the head
        emb = eng.layernorm("norm", x, self.norm, train, rows=B, x_stride=T * D)
        return eng.dropout("head_drop", emb, self.head_drop.p, train)

    def run_backward(self, eng: HipEngine, g_emb: torch.Tensor, on_done=None):
        sv = eng.saved["pe"]
        B, T = sv["B"], sv["T"]
        D = self.num_features
        M = B * T
        a = eng.arena
        gx = eng.scratch("gx0", (M, D))
        hip.zero_(gx)
        g_emb = eng.dropout_backward("head_drop", g_emb, "gemb")
        eng.layernorm_backward("norm", g_emb, gx, T * D)          # rows b*T (class tokens); everything else stays 0
        if on_done is not None:
            on_done(self.norm)
        flip = 1
        for i in range(len(self.blocks) - 1, -1, -1):
            blk = self.blocks[i]
            eng.begin_block(i)
            g2 = eng.dropout_backward(f"b{i}.mlp2_drop", gx, "g2")       # branch gradient; the residual path keeps gx
            if _FUSED_GELU:   # gelu' fused into the fc2 data-gradient epilogue
                d_a = eng.linear_backward_through_gelu(f"b{i}.fc2", f"b{i}.fc1", g2, "da")
            else:
                if "gp" in eng.saved[f"b{i}.act"]:
                    d_a = eng.linear_backward_through_saved_derivative(f"b{i}.fc2", f"b{i}.act", g2, "da")
                else:
                    d_u = eng.dropout_backward(f"b{i}.mlp_drop", eng.linear_backward(f"b{i}.fc2", g2, "du"), "du2")
                    d_a = eng.gelu_backward(f"b{i}.act", d_u, "da")
            d_h = eng.linear_backward(f"b{i}.fc1", d_a, "dh")
            gmid = eng.layernorm_backward(f"b{i}.ln2", d_h, eng.scratch("gmid", (M, D)), D, add=gx)
            d_o = eng.linear_backward(f"b{i}.proj", eng.dropout_backward(f"b{i}.proj_drop", gmid, "g1"), "do")
            d_qkv = eng.attention_backward(f"b{i}.attn", d_o, "dqkv", q_for=f"b{i}.qkv.f8g")
            d_h = eng.linear_backward(f"b{i}.qkv", d_qkv, "dh")
            gx = eng.layernorm_backward(f"b{i}.ln1", d_h, eng.scratch(f"gx{flip}", (M, D)), D, add=gmid)
            flip ^= 1
            eng.end_block(i)
            if on_done is not None:
                on_done(blk)
        eng.begin_block(-1)
        gx = eng.dropout_backward("pos_drop", gx, "gpos")
        # embedding: d_pos = sum_b gx[b], d_cls = sum_b gx[b, 0], d_tok = gx[:, 1:], then the patch projection
        eng.colsum2d(gx, a.grad_flat(self.pos_embed), B, T * D, T * D)
        eng.colsum2d(gx, a.grad_flat(self.cls_token), B, D, T * D)
        npatch = T - 1
        d_tok = eng.scratch("dtok", (B * npatch, D))
        hip.vit_assemble(eng.d, True, d_tok, None, None, gx, B, T, D)
        pr = self.patch_embed.proj
        kp, K = sv["kp"], sv["K"]
        if kp == K:
            eng.wgrad(d_tok, sv["col"], a.grad_flat(pr.weight), N=B * npatch, H=1, W=1, Cin=kp, ldx=kp, P=1,
                           Q=1, Cout=D, lddy=D, dbias=a.grad_flat(pr.bias))
        else:
            dwp = eng.ws.get("pe.dwpad", (D, kp), torch.float32)
            hip.zero_(dwp)
            eng.wgrad(d_tok, sv["col"], dwp, N=B * npatch, H=1, W=1, Cin=kp, ldx=kp, P=1, Q=1, Cout=D, lddy=D)
            hip.add2d(dwp, a.grad_flat(pr.weight), D, K, kp, K)
            eng.colsum2d(d_tok, a.grad_flat(pr.bias), B * npatch, D, D)
        if on_done is not None:
            on_done(self.patch_embed)
            on_done([self.cls_token, self.pos_embed])


_VITS = {
    "vit_base_patch16_224": dict(img=224, patch=16, dim=768, depth=12, heads=12),
    "vit_small_patch16_224": dict(img=224, patch=16, dim=384, depth=12, heads=6),
    "vit_large_patch16_224": dict(img=224, patch=16, dim=1024, depth=24, heads=16),
    "vit_tiny_test": dict(img=64, patch=16, dim=128, depth=2, heads=2),   # reduced member for fast parity tests
    "vit_small_test": dict(img=64, patch=16, dim=256, depth=2, heads=4),  # reduced member inside the fp8 GEMM envelope (dim 256)
}


def create_vit(name: str):
    cfg = _VITS.get(name)
    return HipViT(**cfg) if cfg else None
