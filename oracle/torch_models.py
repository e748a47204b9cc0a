"""ORACLE (test infrastructure, not product code).

Plain torch.nn CPU restatements of the third-party backbones the reference
instantiates through ``timm.create_model(name, pretrained, num_classes=0)``
(/root/reference/nkb_classification/model.py:82,156) plus the reference's two
classifier wrappers (model.py:17-159).  timm itself is NOT vendored under
/root/reference and is not installed here (pyproject.toml:67 leaves it
unpinned), so the topology below is restated from timm's published
``resnet.py`` / ``vision_transformer.py`` and pinned by computed invariants
(parameter counts 11.177 M / 23.508 M / 85.80 M, feature widths 512/2048/768,
state-dict key names) in tests/test_oracle_golden.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Union

import torch
from torch import nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# ResNet (timm layout: stride on the 3x3, downsample = conv1x1(stride)+BN)
# --------------------------------------------------------------------------
class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.act2 = nn.ReLU(inplace=True)
        self.downsample = downsample

    def zero_init_last(self):
        nn.init.zeros_(self.bn2.weight)

    def forward(self, x):
        shortcut = x
        x = self.act1(self.bn1(self.conv1(x)))
        x = self.bn2(self.conv2(x))
        if self.downsample is not None:
            shortcut = self.downsample(shortcut)
        return self.act2(x + shortcut)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.act1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.act2 = nn.ReLU(inplace=True)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.act3 = nn.ReLU(inplace=True)
        self.downsample = downsample

    def zero_init_last(self):
        nn.init.zeros_(self.bn3.weight)

    def forward(self, x):
        shortcut = x
        x = self.act1(self.bn1(self.conv1(x)))
        x = self.act2(self.bn2(self.conv2(x)))
        x = self.bn3(self.conv3(x))
        if self.downsample is not None:
            shortcut = self.downsample(shortcut)
        return self.act3(x + shortcut)


class ResNet(nn.Module):
    """timm.models.resnet.ResNet with num_classes=0 (fc = Identity)."""

    def __init__(self, block, layers: Sequence[int], zero_init_last: bool = True):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.act1 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            stride = 1 if i == 0 else 2
            blocks = []
            for b in range(n):
                s = stride if b == 0 else 1
                down = None
                if s != 1 or inplanes != planes * block.expansion:
                    down = nn.Sequential(
                        nn.Conv2d(inplanes, planes * block.expansion, 1, s, bias=False),
                        nn.BatchNorm2d(planes * block.expansion),
                    )
                blocks.append(block(inplanes, planes, s, down))
                inplanes = planes * block.expansion
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.num_features = inplanes
        self.fc = nn.Identity()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if zero_init_last:
            for m in self.modules():
                if hasattr(m, "zero_init_last"):
                    m.zero_init_last()

    def forward_features(self, x):
        x = self.maxpool(self.act1(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def forward(self, x):
        x = self.forward_features(x)
        return self.fc(x.mean((2, 3)))


# --------------------------------------------------------------------------
# timm VisionTransformer (vit_base_patch16_224, num_classes=0, token pooling)
# --------------------------------------------------------------------------
class PatchEmbed(nn.Module):
    def __init__(self, patch: int, in_chans: int, dim: int):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, patch, patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class Attention(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.num_heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3)
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)

    def forward(self, x):
        B, T, D = x.shape
        qkv = self.qkv(x).reshape(B, T, 3, self.num_heads, D // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (q * self.scale) @ k.transpose(-2, -1)
        att = self.attn_drop(att.softmax(dim=-1))
        x = (att @ v).transpose(1, 2).reshape(B, T, D)
        return self.proj_drop(self.proj(x))


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.drop1 = nn.Dropout(0.0)
        self.fc2 = nn.Linear(hidden, dim)
        self.drop2 = nn.Dropout(0.0)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class Block(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_ratio: float):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class VisionTransformer(nn.Module):
    def __init__(self, img=224, patch=16, dim=768, depth=12, heads=12, mlp_ratio=4.0):
        super().__init__()
        self.num_features = dim
        self.patch_embed = PatchEmbed(patch, 3, dim)
        n_tok = (img // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n_tok + 1, dim) * 0.02)
        self.pos_drop = nn.Dropout(0.0)
        self.blocks = nn.Sequential(*[Block(dim, heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head_drop = nn.Dropout(0.0)      # timm forward_head: fc_norm -> head_drop -> head (Identity for num_classes=0)
        self.head = nn.Identity()
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        x = self.pos_drop(x + self.pos_embed)
        x = self.norm(self.blocks(x))
        return self.head(self.head_drop(x[:, 0]))


# --------------------------------------------------------------------------
# unicom VisionTransformer (model.py:77-79: unicom.load(name.split()[1])[0]).
# PARITY UNPINNED: deepglint/unicom is an un-pinned git dependency
# (pyproject.toml:81), absent from /root/reference and not installed; this
# restates its published vision_transformer.py as recorded in SURVEY.md §8 A9
# (no class token, bias-free qkv, fp32 softmax, ReLU6 MLP, DropPath,
# LayerNorm eps 1e-5, `feature` = Linear -> BN1d(2e-5) -> Linear -> BN1d(2e-5)).
# Pinned only by computed invariants (572.33 M parameters for ViT-L/14, the
# 768-wide feature[-2], key names) in tests/test_oracle_golden.py.
# --------------------------------------------------------------------------
def _at_least_f32(t: torch.Tensor) -> torch.Tensor:
    """The package's `.float()` islands (softmax, final norm): fp32 under autocast, a no-op for the float64 truth runs."""
    return t if t.dtype == torch.float64 else t.float()


class UnicomAttention(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.num_heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, T, D = x.shape
        qkv = self.qkv(x).reshape(B, T, 3, self.num_heads, D // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = ((_at_least_f32(q) @ _at_least_f32(k).transpose(-2, -1)) * self.scale).softmax(dim=-1)   # fp32 island
        x = (att @ _at_least_f32(v)).to(x.dtype).transpose(1, 2).reshape(B, T, D)
        return self.proj(x)


class UnicomMlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.ReLU6()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class DropPath(nn.Module):
    """Per-sample stochastic depth; `forced_keep` ([B] of 0/1) replays a recorded draw in the parity tests."""

    def __init__(self, drop_prob: float):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.forced_keep = None

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep_prob = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.dim() - 1)
        keep = (self.forced_keep.to(x.dtype).reshape(shape) if self.forced_keep is not None
                else torch.bernoulli(torch.full(shape, keep_prob, dtype=x.dtype)))
        return x * keep / keep_prob


class UnicomBlock(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_ratio: int, drop_path: float):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = UnicomAttention(dim, heads)
        self.drop_path = DropPath(drop_path) if drop_path > 0 else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = UnicomMlp(dim, dim * mlp_ratio)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class UnicomPatchEmbedding(nn.Module):
    def __init__(self, img: int, patch: int, in_chans: int, dim: int):
        super().__init__()
        self.num_patches = (img // patch) ** 2
        self.proj = nn.Conv2d(in_chans, dim, patch, patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class UnicomViT(nn.Module):
    def __init__(self, input_size=224, patch_size=32, dim=768, embedding_size=768, depth=12, num_heads=12,
                 drop_path_rate=0.1, mlp_ratio=4):
        super().__init__()
        self.dim = dim
        self.patch_embed = UnicomPatchEmbedding(input_size, patch_size, 3, dim)
        T = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, T, dim))
        self.blocks = nn.ModuleList([UnicomBlock(dim, num_heads, mlp_ratio, drop_path_rate) for _ in range(depth)])
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

    def forward(self, x):
        B = x.shape[0]
        x = self.patch_embed(x) + self.pos_embed
        for blk in self.blocks:
            x = blk(x)
        x = self.norm(_at_least_f32(x))
        return self.feature(x.reshape(B, -1))


_UNICOM = {
    "vit-b/32": dict(input_size=224, patch_size=32, dim=768, embedding_size=512, depth=12, num_heads=12),
    "vit-b/16": dict(input_size=224, patch_size=16, dim=768, embedding_size=768, depth=12, num_heads=12),
    "vit-l/14": dict(input_size=224, patch_size=14, dim=1024, embedding_size=768, depth=24, num_heads=16),
    "vit-l/14@336px": dict(input_size=336, patch_size=14, dim=1024, embedding_size=768, depth=24, num_heads=16),
    "vit-tiny-test": dict(input_size=56, patch_size=14, dim=128, embedding_size=64, depth=2, num_heads=2),
    "vit-small-test": dict(input_size=56, patch_size=14, dim=256, embedding_size=64, depth=2, num_heads=4),
}


_BACKBONES = {
    "resnet18": lambda: ResNet(BasicBlock, (2, 2, 2, 2)),
    "resnet34": lambda: ResNet(BasicBlock, (3, 4, 6, 3)),
    "resnet50": lambda: ResNet(Bottleneck, (3, 4, 6, 3)),
    "resnet101": lambda: ResNet(Bottleneck, (3, 4, 23, 3)),
    "vit_base_patch16_224": lambda: VisionTransformer(),
    # reduced-size members of the same families, used by fast parity tests
    "resnet_tiny_basic": lambda: ResNet(BasicBlock, (1, 1, 1, 1)),
    "resnet_tiny_bottleneck": lambda: ResNet(Bottleneck, (1, 1, 1, 1)),
    "vit_tiny_test": lambda: VisionTransformer(img=64, patch=16, dim=128, depth=2, heads=2),
    "vit_small_test": lambda: VisionTransformer(img=64, patch=16, dim=256, depth=2, heads=4),
}


def create_backbone(name: str) -> nn.Module:
    """Stand-in for ``timm.create_model(name, pretrained=False, num_classes=0)``."""
    if name.lower().startswith("unicom"):        # model.py:77-79
        return UnicomViT(**_UNICOM[name.split()[1].lower()])
    if name not in _BACKBONES:
        raise NotImplementedError(f"oracle has no restatement of backbone {name!r}")
    return _BACKBONES[name]()


# --------------------------------------------------------------------------
# Classifier wrappers (reference model.py:17-159), restated
# --------------------------------------------------------------------------
def _init_head(params, strategy: str):
    # model.py:45-57 — only the kaiming_* strategies run at reference HEAD
    for p in params:
        if p.ndim >= 2:
            if strategy == "kaiming_normal_":
                nn.init.kaiming_normal_(p, nonlinearity="relu")
            elif strategy == "kaiming_uniform_":
                nn.init.kaiming_uniform_(p, nonlinearity="relu")
            else:
                raise TypeError(f"classifier_initialization {strategy!r} fails in the reference (model.py:52-55)")
        else:
            nn.init.zeros_(p)


def _set_dropout(model: nn.Module, p: float):
    # model.py:66-72
    for child in model.children():
        if isinstance(child, nn.Dropout):
            child.p = p
        _set_dropout(child, p)


class OracleClassifier(nn.Module):
    """Single- (list classes) or multi-task (dict classes) classifier on CPU."""

    def __init__(self, cfg_model: dict, classes: Union[List, Dict[str, List]]):
        super().__init__()
        self.emb_model = create_backbone(cfg_model["model"])
        # model.py:79 reads feature[-2].out_features for unicom, model.py:83 num_features for timm: both are this
        self.emb_size = self.emb_model.num_features
        _set_dropout(self.emb_model, cfg_model.get("backbone_dropout", 0.0))
        pd = cfg_model.get("classifier_dropout", 0.0)
        self.multi = isinstance(classes, dict)
        if self.multi:
            self.classifier = nn.ModuleDict(
                {t: nn.Sequential(nn.Dropout(pd), nn.Linear(self.emb_size, len(c))) for t, c in classes.items()}
            )
        else:
            self.classifier = nn.Sequential(nn.Dropout(pd), nn.Linear(self.emb_size, len(classes)))
        _init_head(self.classifier.parameters(), cfg_model.get("classifier_initialization", "kaiming_normal_"))

    def set_backbone_state(self, state: str):
        for p in self.emb_model.parameters():
            if state == "freeze":
                p.requires_grad = False
            elif state == "unfreeze":
                p.requires_grad = True

    def forward(self, x):
        emb = self.emb_model(x)
        if self.multi:
            return {t: head(emb) for t, head in self.classifier.items()}
        return self.classifier(emb)


def count_params(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())
