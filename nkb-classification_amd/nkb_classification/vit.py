"""timm VisionTransformer family for the HIP engine (filled in once the ResNet path is parity-green)."""
from __future__ import annotations


def create_vit(name: str):
    return None
