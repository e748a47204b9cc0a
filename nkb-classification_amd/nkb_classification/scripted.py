"""TorchScript export of a trained HIP model — the `scripted_{best,last}.pt` archives of the reference's train loop
(/root/reference/train.py:66-73: `torch.jit.script(model)` saved next to every state-dict checkpoint).

The HIP classifier is one custom autograd node over a flat parameter arena, not a torch module graph, so there is
nothing for `torch.jit.script` to trace.  What the reference's downstream tools (`inference.py`, `export.py`,
`get_model(..., scripted=True)`, model.py:163-165) need from the archive is a self-contained module with the same
`forward` contract and the trained weights; this file builds exactly that: a plain `torch.nn` module with timm's
parameter names (so `load_state_dict(hip_model.state_dict())` is exact) which torch scripts and runs with its own
kernels on any device.  It is an export artefact only — never used by `train_epoch` / `val_epoch`.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn


class _BasicBlock(nn.Module):
    def __init__(self, inplanes: int, planes: int, stride: int, downsample: Optional[nn.Module]):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample if downsample is not None else nn.Identity()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = torch.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return torch.relu(y + self.downsample(x))


class _Bottleneck(nn.Module):
    def __init__(self, inplanes: int, planes: int, stride: int, downsample: Optional[nn.Module]):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample if downsample is not None else nn.Identity()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = torch.relu(self.bn1(self.conv1(x)))
        y = torch.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return torch.relu(y + self.downsample(x))


class _ResNet(nn.Module):
    def __init__(self, bottleneck: bool, layers):
        super().__init__()
        exp = 4 if bottleneck else 1
        block = _Bottleneck if bottleneck else _BasicBlock
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes = 64
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            blocks: List[nn.Module] = []
            for b in range(n):
                stride = 2 if (b == 0 and li > 0) else 1
                ds = None
                if b == 0 and (stride != 1 or inplanes != planes * exp):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * exp, 1, stride, bias=False), nn.BatchNorm2d(planes * exp))
                blocks.append(block(inplanes, planes, stride, ds))
                inplanes = planes * exp
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
        self.num_features = inplanes

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.maxpool(torch.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return x.mean((2, 3))


class _Attention(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, D = x.shape
        qkv = self.qkv(x).reshape(B, T, 3, self.num_heads, D // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (q @ k.transpose(-2, -1)) * (float(D // self.num_heads) ** -0.5)
        y = (att.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
        return self.proj(y)


class _Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc2(torch.nn.functional.gelu(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim: int, heads: int, mlp_ratio: float):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(self, patch: int, dim: int):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, patch)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.proj(x).flatten(2).transpose(1, 2)


class _ViT(nn.Module):
    def __init__(self, img: int, patch: int, dim: int, depth: int, heads: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.num_features = dim
        self.patch_embed = _PatchEmbed(patch, dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, (img // patch) ** 2 + 1, dim))
        self.blocks = nn.Sequential(*[_Block(dim, heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1) + self.pos_embed
        return self.norm(self.blocks(x))[:, 0]


class _UnicomAttention(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, D = x.shape
        qkv = self.qkv(x).reshape(B, T, 3, self.num_heads, D // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (q @ k.transpose(-2, -1)) * (float(D // self.num_heads) ** -0.5)
        y = (att.softmax(dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
        return self.proj(y)


class _UnicomMlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc2(torch.nn.functional.relu6(self.fc1(x)))


class _UnicomBlock(nn.Module):
    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _UnicomAttention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _UnicomMlp(dim, dim * 4)

    def forward(self, x: torch.Tensor) -> torch.Tensor:      # stochastic depth is the identity at inference
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _UnicomViT(nn.Module):
    def __init__(self, img: int, patch: int, dim: int, emb: int, depth: int, heads: int):
        super().__init__()
        self.patch_embed = _PatchEmbed(patch, dim)
        T = (img // patch) ** 2
        self.pos_embed = nn.Parameter(torch.zeros(1, T, dim))
        self.blocks = nn.Sequential(*[_UnicomBlock(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)
        self.feature = nn.Sequential(nn.Linear(dim * T, dim, bias=False), nn.BatchNorm1d(dim, eps=2e-5),
                                     nn.Linear(dim, emb, bias=False), nn.BatchNorm1d(emb, eps=2e-5))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.norm(self.blocks(self.patch_embed(x) + self.pos_embed))
        return self.feature(x.reshape(x.shape[0], -1))


class _SingleHead(nn.Module):
    def __init__(self, emb_model: nn.Module, emb: int, n: int):
        super().__init__()
        self.emb_model = emb_model
        self.classifier = nn.Sequential(nn.Dropout(0.0), nn.Linear(emb, n))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.classifier(self.emb_model(x))


class _MultiHead(nn.Module):
    def __init__(self, emb_model: nn.Module, emb: int, sizes: Dict[str, int]):
        super().__init__()
        self.emb_model = emb_model
        self.classifier = nn.ModuleDict({t: nn.Sequential(nn.Dropout(0.0), nn.Linear(emb, n)) for t, n in sizes.items()})

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        emb = self.emb_model(x)
        out: Dict[str, torch.Tensor] = {}
        for name, head in self.classifier.items():
            out[name] = head(emb)
        return out


def _backbone_like(hip_backbone) -> nn.Module:
    if getattr(hip_backbone, "family", "") == "unicom":
        return _UnicomViT(hip_backbone.img, hip_backbone.patch, hip_backbone.dim, hip_backbone.num_features,
                          len(hip_backbone.blocks), hip_backbone.heads)
    if getattr(hip_backbone, "family", "") == "vit":
        return _ViT(hip_backbone.img, hip_backbone.patch, hip_backbone.num_features, len(hip_backbone.blocks), hip_backbone.heads)
    layers = [len(getattr(hip_backbone, f"layer{i}")) for i in (1, 2, 3, 4)]
    bottleneck = hasattr(getattr(hip_backbone, "layer1")[0], "conv3")
    return _ResNet(bottleneck, layers)


def build_scriptable(hip_model) -> nn.Module:
    """Plain-torch module with the HIP model's architecture, weights and forward contract (eval mode, CPU)."""
    emb = _backbone_like(hip_model.emb_model)
    if isinstance(hip_model.classifier, nn.ModuleDict):
        twin = _MultiHead(emb, hip_model.emb_size, {t: h[1].out_features for t, h in hip_model.classifier.items()})
    else:
        twin = _SingleHead(emb, hip_model.emb_size, hip_model.classifier[1].out_features)
    state = {k: v.detach().to("cpu") for k, v in hip_model.state_dict().items()}
    twin.load_state_dict(state, strict=True)
    return twin.eval()


def save_scripted(hip_model, path) -> None:
    """`torch.jit.script` archive of the trained model, as train.py:66-73 writes per epoch."""
    torch.jit.script(build_scriptable(hip_model)).save(str(path))
