"""unicom VisionTransformer family ("unicom ViT-B/32", "unicom ViT-B/16", "unicom ViT-L/14", "unicom ViT-L/14@336px")
for the HIP engine: parameter containers with the unicom state-dict names + the forward / backward execution plan.

Reference call site: `unicom.load(name.split()[1])[0]` at /root/reference/nkb_classification/model.py:77-79, with
`emb_size = emb_model.feature[-2].out_features`.  The unicom package (deepglint/unicom, un-pinned git dependency,
pyproject.toml:81) is not vendored under /root/reference and not installed, so the architecture below restates its
published `vision_transformer.py` as recorded in SURVEY.md §8 A9 — PARITY UNPINNED: it is checked against this repo's
own torch-CPU restatement (oracle/torch_models.py: UnicomViT), not against outputs of the package itself.

  patch_embed.proj  Conv2d(3, D, p, p)            -> [B, T, D]   (T = (img / p)^2, no class token)
  + pos_embed [1, T, D]
  depth x { x += drop_path(attn(norm1(x)));  x += drop_path(mlp(norm2(x))) }
      norm*: LayerNorm(D) (eps 1e-5);  attn: qkv Linear(D, 3D, bias=False), softmax in fp32, proj Linear(D, D);
      mlp: Linear(D, 4D) -> ReLU6 -> Linear(4D, D);  drop_path: per-sample stochastic depth (rate 0.1)
  norm LayerNorm(D) over every token, flatten to [B, T*D]
  feature: Linear(T*D, D, bias=False) -> BatchNorm1d(D, eps 2e-5) -> Linear(D, E, bias=False) -> BatchNorm1d(E, eps 2e-5)

The package wraps each block in activation checkpointing (a memory optimisation, numerically the identity); with 288 GB
of HBM the activations of the B=128 configuration are simply kept.
"""
from __future__ import annotations

import os

import torch
from torch import nn

from . import hip
from .backbones import _ParamOnly
from .hipnet import HipEngine

_FUSED_RELU6 = True   # ReLU6 in the fc1 epilogue, its mask in the fc2 data gradient


class _PatchEmbedding(_ParamOnly):
    def __init__(self, img, patch, in_chans, dim):
        super().__init__()
        self.num_patches = (img // patch) ** 2
        self.proj = nn.Conv2d(in_chans, dim, patch, patch)


class _Attention(_ParamOnly):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)


class _Mlp(_ParamOnly):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.ReLU6()
        self.fc2 = nn.Linear(hidden, dim)


class _DropPath(_ParamOnly):
    def __init__(self, drop_prob: float):
        super().__init__()
        self.drop_prob = float(drop_prob)


class _Block(_ParamOnly):
    def __init__(self, dim, heads, mlp_ratio, drop_path):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Attention(dim, heads)
        self.drop_path = _DropPath(drop_path) if drop_path > 0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, dim * mlp_ratio)


class HipUnicomViT(_ParamOnly):
    family = "unicom"

    def __init__(self, input_size=224, patch_size=32, dim=768, embedding_size=768, depth=12, num_heads=12,
                 drop_path_rate=0.1, mlp_ratio=4):
        super().__init__()
        self.dim, self.img, self.patch, self.heads = dim, input_size, patch_size, num_heads
        self.patch_embed = _PatchEmbedding(input_size, patch_size, 3, dim)
        T = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, T, dim))
        self.blocks = nn.ModuleList([_Block(dim, num_heads, mlp_ratio, drop_path_rate) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)
        self.feature = nn.Sequential(
            nn.Linear(dim * T, dim, bias=False),
            nn.BatchNorm1d(dim, eps=2e-5),
            nn.Linear(dim, embedding_size, bias=False),
            nn.BatchNorm1d(embedding_size, eps=2e-5),
        )
        self.num_features = embedding_size
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def gemm_convs(self):
        return [m for m in self.modules() if isinstance(m, nn.Linear)]

    def stem_convs(self):
        return [self.patch_embed.proj]

    def fp8_linears(self):
        """The Linear layers of the transformer blocks: the contractions that cfg.amp_dtype = "fp8" moves to fp8 operands."""
        return [m for blk in self.blocks for m in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2)]

    @staticmethod
    def _dp(blk) -> float:
        return blk.drop_path.drop_prob if isinstance(blk.drop_path, _DropPath) else 0.0

    def run_forward(self, eng: HipEngine, img: torch.Tensor, train: bool) -> torch.Tensor:
        for k in [k for k in eng.saved if k.endswith(".dp1") or k.endswith(".dp2")]:
            del eng.saved[k]                                 # stochastic-depth draws of a previous step
        B, C, Hh, Ww = img.shape
        if Hh != self.img or Ww != self.img:
            raise RuntimeError(f"this ViT expects {self.img}x{self.img} inputs (pos_embed is fixed), got {Hh}x{Ww}")
        pr = self.patch_embed.proj
        D, ps, T = self.dim, self.patch, self.patch_embed.num_patches
        K = C * ps * ps
        kp = eng.kpad(K)
        a = eng.arena
        col = eng.ws.get("pe.col", (B * T, kp), eng.T)
        hip.im2row(eng.d, img, col, B, C, Hh, Ww, ps, ps, ps, 0, kp)
        tok = eng.ws.get("pe.tok", (B * T, D), eng.T)
        hip.conv_gemm(eng.d, 0, col, eng.w_fwd(pr.weight), tok, N=B * T, H=1, W=1, Cin=kp, ldx=kp, P=1, Q=1, Cout=D,
                      ldy=D, bias=a.param_flat(pr.bias))
        x = eng.ws.get("pe.x", (B * T, D), eng.T)
        hip.vit_assemble(eng.d, False, tok, None, a.param_flat(self.pos_embed), x, B, T, D)
        if train:
            eng.saved["pe"] = dict(col=col, B=B, T=T, kp=kp, K=K)
        for i, blk in enumerate(self.blocks):
            at, mlp = blk.attn, blk.mlp
            dp = self._dp(blk) if train else 0.0
            h = eng.layernorm(f"b{i}.ln1", x, blk.norm1, train, q_for=f"b{i}.qkv.f8x")
            qkv = eng.linear(f"b{i}.qkv", h, at.qkv, train)
            o = eng.attention(f"b{i}.attn", qkv, B, T, self.heads, train, q_for=f"b{i}.proj.f8x")
            if dp > 0:     # y = x + keep[b] / (1 - p) * proj(o): the per-sample scale rides in the GEMM call
                x = eng.linear(f"b{i}.proj", o, at.proj, train, add=x, row_scale=(eng.drop_path_scale(f"b{i}.dp1", dp, B), T))
            else:
                x = eng.linear(f"b{i}.proj", o, at.proj, train, add=x)
            h = eng.layernorm(f"b{i}.ln2", x, blk.norm2, train, q_for=f"b{i}.fc1.f8x")
            if _FUSED_RELU6:
                u = eng.linear_relu6(f"b{i}.fc1", h, mlp.fc1, train, q_for=f"b{i}.fc2.f8x", consumer=mlp.fc2)
            else:
                u = eng.relu6(f"b{i}.act", eng.linear(f"b{i}.fc1", h, mlp.fc1, train), train)
            if dp > 0:
                x = eng.linear(f"b{i}.fc2", u, mlp.fc2, train, add=x, row_scale=(eng.drop_path_scale(f"b{i}.dp2", dp, B), T))
            else:
                x = eng.linear(f"b{i}.fc2", u, mlp.fc2, train, add=x)
        y = eng.layernorm("norm", x, self.norm, train)                      # every token feeds the feature head
        f = self.feature
        e1 = eng.conv_bn("feat1", y.view(B, T * D), f[0], f[1], False, None, train)
        e2 = eng.conv_bn("feat2", e1.view(B, D), f[2], f[3], False, None, train)
        return e2.view(B, self.num_features)

    def run_backward(self, eng: HipEngine, g_emb: torch.Tensor, on_done=None):
        sv = eng.saved["pe"]
        B, T = sv["B"], sv["T"]
        D = self.dim
        M = B * T
        a = eng.arena
        f = self.feature
        eng.begin_block(len(self.blocks))
        g = eng.bn_backward("feat2", g_emb.reshape(B, 1, 1, self.num_features), "gf2")
        g = eng.conv_backward("feat2", g, "gf2x")
        g = eng.bn_backward("feat1", g, "gf1")
        g = eng.conv_backward("feat1", g, "gf1x")                             # [B,1,1,T*D]
        nb = len(self.blocks)
        gx = eng.layernorm_backward("norm", g.view(M, D), eng.scratch("gx0", (M, D)), D,
                                    consumer=f"b{nb - 1}.fc2" if nb else None, consumer_dp=f"b{nb - 1}.dp2", consumer_block=nb - 1)
        eng.end_block(len(self.blocks))
        if on_done is not None:
            on_done(f)
            on_done(self.norm)
        flip = 1
        for i in range(len(self.blocks) - 1, -1, -1):
            blk = self.blocks[i]
            eng.begin_block(i)
            gs2 = eng.drop_path_gscale(f"b{i}.dp2", M)                       # branch gradient = scale * gx; the residual path keeps gx
            if _FUSED_RELU6:
                d_a = eng.linear_backward_through_relu6(f"b{i}.fc2", f"b{i}.fc1", gx, "da", q_for=f"b{i}.fc1.f8g", g_scale=gs2)
            else:
                d_u = eng.linear_backward(f"b{i}.fc2", gx, "du", g_scale=gs2)
                d_a = eng.relu6_backward(f"b{i}.act", d_u, "da")
            d_h = eng.linear_backward(f"b{i}.fc1", d_a, "dh")
            gmid = eng.layernorm_backward(f"b{i}.ln2", d_h, eng.scratch("gmid", (M, D)), D, add=gx,
                                          consumer=f"b{i}.proj", consumer_dp=f"b{i}.dp1")
            d_o = eng.linear_backward(f"b{i}.proj", gmid, "do", g_scale=eng.drop_path_gscale(f"b{i}.dp1", M))
            d_qkv = eng.attention_backward(f"b{i}.attn", d_o, "dqkv", q_for=f"b{i}.qkv.f8g")
            d_h = eng.linear_backward(f"b{i}.qkv", d_qkv, "dh")
            gx = eng.layernorm_backward(f"b{i}.ln1", d_h, eng.scratch(f"gx{flip}", (M, D)), D, add=gmid,
                                        consumer=f"b{i - 1}.fc2" if i > 0 else None, consumer_dp=f"b{i - 1}.dp2", consumer_block=i - 1)
            flip ^= 1
            eng.end_block(i)
            if on_done is not None:
                on_done(blk)
        eng.begin_block(-1)
        # embedding: d_pos = sum_b gx[b]; the token gradient is gx itself (no class token)
        eng.colsum2d(gx, a.grad_flat(self.pos_embed), B, T * D, T * D)
        pr = self.patch_embed.proj
        kp, K = sv["kp"], sv["K"]
        if kp == K:
            eng.wgrad(gx, sv["col"], a.grad_flat(pr.weight), N=M, H=1, W=1, Cin=kp, ldx=kp, P=1, Q=1, Cout=D,
                           lddy=D, dbias=a.grad_flat(pr.bias))
        else:
            dwp = eng.ws.get("pe.dwpad", (D, kp), torch.float32)
            hip.zero_(dwp)
            eng.wgrad(gx, sv["col"], dwp, N=M, H=1, W=1, Cin=kp, ldx=kp, P=1, Q=1, Cout=D, lddy=D)
            hip.add2d(dwp, a.grad_flat(pr.weight), D, K, kp, K)
            eng.colsum2d(gx, a.grad_flat(pr.bias), M, D, D)
        if on_done is not None:
            on_done(self.patch_embed)
            on_done([self.pos_embed])


# unicom.load(name) -> build_model(name): the four published members (embedding_size = output width of `feature`)
_UNICOM = {
    "vit-b/32": dict(input_size=224, patch_size=32, dim=768, embedding_size=512, depth=12, num_heads=12),
    "vit-b/16": dict(input_size=224, patch_size=16, dim=768, embedding_size=768, depth=12, num_heads=12),
    "vit-l/14": dict(input_size=224, patch_size=14, dim=1024, embedding_size=768, depth=24, num_heads=16),
    "vit-l/14@336px": dict(input_size=336, patch_size=14, dim=1024, embedding_size=768, depth=24, num_heads=16),
    # reduced member for the fast parity tests (same blocks, 16 tokens)
    "vit-tiny-test": dict(input_size=56, patch_size=14, dim=128, embedding_size=64, depth=2, num_heads=2),
    # ... and one whose Linear layers are inside the fp8 GEMM envelope (K >= 256, N % 256 == 0)
    "vit-small-test": dict(input_size=56, patch_size=14, dim=256, embedding_size=64, depth=2, num_heads=4),
}


def load(name: str):
    """Counterpart of `unicom.load(name)[0]` (model.py:78): the architecture with fresh weights — the package's weight
    hub is unreachable offline; trained unicom weights come in through cfg.model['checkpoint'] (same key names)."""
    cfg = _UNICOM.get(name.lower())
    if cfg is None:
        raise RuntimeError(f"Model {name} not found; available models = {[k for k in _UNICOM if 'test' not in k]}")
    return HipUnicomViT(**cfg)
