"""Backbone definitions for the HIP engine.

The reference builds its backbone with `timm.create_model(name, pretrained=..., num_classes=0)`
(/root/reference/nkb_classification/model.py:82).  timm is a third-party dependency that is not part of the
reference tree; the modules below keep timm's parameter names and shapes (so `state_dict()` checkpoints are
interchangeable, train.py:70-72 / model.py:172) but hold parameters only — their arithmetic is executed by
HipEngine through libnkbhip, never by torch.nn forward methods.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
from torch import nn

from .hipnet import HipEngine

# bn1 -> relu -> maxpool of the stem as one kernel each way (0 = separate kernels, for A/B measurements)
_FUSED_STEM_TAIL = True
# forward: projection-shortcut convolution on the side stream, next to the block's main branch
_SIDE_SHORTCUT = True


class _ParamOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guard
        raise RuntimeError("this module only stores parameters; it is executed by the owning HIP classifier "
                           "(call the classifier returned by get_model, on a cuda device)")


class _Basic(_ParamOnly):
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def stages(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2)]

    def last_bn(self):
        return self.bn2


class _Bottle(_ParamOnly):
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def stages(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]

    def last_bn(self):
        return self.bn3


class HipResNet(_ParamOnly):
    """timm `resnet*` parameter layout: conv1/bn1, layer1..4.{i}.conv{j}/bn{j}/downsample.{0,1}."""

    family = "resnet"

    def __init__(self, block, layers: Sequence[int], zero_init_last: bool = True):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            stride = 1 if i == 0 else 2
            blocks = []
            for b in range(n):
                s = stride if b == 0 else 1
                down = None
                if s != 1 or inplanes != planes * block.expansion:
                    down = nn.Sequential(nn.Conv2d(inplanes, planes * block.expansion, 1, s, bias=False),
                                         nn.BatchNorm2d(planes * block.expansion))
                blocks.append(block(inplanes, planes, s, down))
                inplanes = planes * block.expansion
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.num_features = inplanes
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if zero_init_last:
            for m in self.modules():
                if isinstance(m, (_Basic, _Bottle)):
                    nn.init.zeros_(m.last_bn().weight)

    def blocks(self):
        for i in range(1, 5):
            for j, blk in enumerate(getattr(self, f"layer{i}")):
                yield f"layer{i}.{j}", blk

    def gemm_convs(self) -> List[nn.Conv2d]:
        """Convolutions executed by the generic implicit-GEMM kernel (everything but the im2row stem)."""
        out = []
        for _, blk in self.blocks():
            out += [c for c, _ in blk.stages()]
            if blk.downsample is not None:
                out.append(blk.downsample[0])
        return out

    def stem_convs(self) -> List[nn.Conv2d]:
        return [self.conv1]

    # ---- execution plan ---------------------------------------------------------------
    def run_forward(self, eng: HipEngine, img: torch.Tensor, train: bool) -> torch.Tensor:
        from . import hip
        N, C, H, W = img.shape
        conv = self.conv1
        R, st, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        P, Q = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
        fused = _FUSED_STEM_TAIL
        if eng.packed_stem(conv):
            xp = eng.ws.get("stem.xp", (N, H, (W + 1) // 2 * 2, 4), eng.T)
            hip.stem_pack(eng.d, img, xp, N, C, H, W)
            x = eng.conv_bn("stem", xp, conv, self.bn1, True, None, train, pool=fused, stem_packed=(N, H, W))
        else:
            kp = eng.kpad(C * R * R)
            col = eng.ws.get("stem.col", (N, P, Q, kp), eng.T)
            hip.im2row(eng.d, img, col, N, C, H, W, R, R, st, pad, kp)
            x = eng.conv_bn("stem", col, conv, self.bn1, True, None, train, col_input=True, pool=fused)
        if not fused:
            x = eng.maxpool("pool", x, train)
        all_blocks = list(self.blocks())
        nblocks = len(all_blocks)
        for bi, (name, blk) in enumerate(all_blocks):
            inp = x
            stages = blk.stages()
            short, short_affine = inp, None
            # the Gram form's backward needs the masked output gradient WITH its sums from the next block's first data gradient
            # (HipEngine.can_fuse_residual_bn_backward): decide on that producer here, where the forward form is chosen — a next
            # block whose first convolution takes the parity-class data gradient (3x3 / stride 2) cannot supply them
            want_gram = (isinstance(blk, _Bottle) and bi + 1 < nblocks and train
                         and not eng.s2_classes(all_blocks[bi + 1][1].stages()[0][0]))
            proj = None
            if blk.downsample is not None and want_gram and eng.gram_proj_ok(blk.downsample[0], stages[-1][0]):
                proj = (blk.downsample[0], blk.downsample[1], inp, f"{name}.ds")     # rides inside the closing convolution
            elif blk.downsample is not None:
                # projection shortcut: its BatchNorm is applied inside the closing stage's pass (never materialised);
                # the convolution itself only depends on the block input, so it runs on the side stream next to the
                # main branch (the forward pass has nothing else to overlap)
                box = {}

                def shortcut(key=f"{name}.ds", src=inp, ds=blk.downsample):
                    box["r"] = eng.conv_bn(key, src, ds[0], ds[1], False, None, train, defer_apply=True)
                if _SIDE_SHORTCUT:
                    eng.on_side(shortcut)
                else:
                    shortcut()
            # Gram form of the closing stage (hipnet._conv_bn_gram): every bottleneck but the last — its backward needs the masked
            # output gradient + sums that the NEXT block's conv1 data gradient leaves (can_fuse_residual_bn_backward)
            for k, (cv, bn) in enumerate(stages[:-1]):
                x = eng.conv_bn(f"{name}.{k}", x, cv, bn, True, None, train, gram_out=want_gram and k == len(stages) - 2)
            if blk.downsample is not None and proj is None:
                if _SIDE_SHORTCUT:
                    eng.join_side()
                short, s_scale, s_shift = box["r"]
                short_affine = (s_scale, s_shift) if s_scale is not None else None     # None: eval mode, already normalised
            cv, bn = stages[-1]
            gram = want_gram and eng.can_fuse_bn_backward(f"{name}.{len(stages) - 2}")
            if proj is not None and not gram:
                raise RuntimeError(f"{name}: projection shortcut planned into a Gram-form closing stage that did not qualify")
            x = eng.conv_bn(f"{name}.{len(stages) - 1}", x, cv, bn, True, short, train, res_affine=short_affine, gram=gram, proj=proj)
        return eng.avgpool("gap", x)

    def run_backward(self, eng: HipEngine, g_emb: torch.Tensor, on_done=None):
        """on_done(module) is called as soon as every parameter gradient of `module` (layer4 .. layer1, then the
        stem) is final, so the data-parallel reducer can start exchanging it while backward continues."""
        g = eng.avgpool_backward("gap", g_emb, "g0")
        flip = 1
        blocks = list(self.blocks())
        g_stats = None        # set when g arrives already masked, with its BN reduction done by the producing dgrad
        for bi in range(len(blocks) - 1, -1, -1):
            name, blk = blocks[bi]
            eng.begin_block(bi)
            n = len(blk.stages())
            # last stage: ReLU(bn(conv) + shortcut); the masked incoming gradient also feeds the shortcut
            last = f"{name}.{n - 1}"
            top = n - 1
            if eng.saved[last].get("gram") is not None:
                # Gram-form closing stage: no BN-backward apply pass, no gradient of the raw conv output — R = g^T a, the small
                # algebra, and one data gradient over [g | a] that lands directly on the stage before
                if g_stats is None:
                    raise RuntimeError(f"{last}: the Gram-form closing stage needs the fused residual BN-backward producer")
                prev = f"{name}.{n - 2}"
                ga, st = eng.gram_closing_backward(last, g, g_stats, prev, f"a{n - 1}")
                gc = eng.bn_backward_fused(prev, ga, st, f"c{n - 1}")
                bits = None
                top = n - 2
            elif g_stats is not None:
                gc = eng.bn_backward_fused(last, g, g_stats, "t0")
                bits = None                              # g is masked already
            else:
                gc = eng.bn_backward(last, g, "t0", write_masked=True)
                bits = eng.saved[last].get("bits")       # set: g was NOT masked in place, consumers apply the bits
            for k in range(top, 0, -1):
                prev = f"{name}.{k - 1}"
                if eng.can_fuse_bn_backward(prev):
                    ga, st = eng.conv_backward(f"{name}.{k}", gc, f"a{k}", fuse_bn=prev)
                    gc = eng.bn_backward_fused(prev, ga, st, f"c{k}")
                else:
                    ga = eng.conv_backward(f"{name}.{k}", gc, f"a{k}")
                    gc = eng.bn_backward(prev, ga, f"c{k}")
            add, add_hw, add_bits = g, (0, 0), bits
            if blk.downsample is not None and eng.saved[last].get("gram_ds") is not None:
                add, add_bits, eng._gram_ds_grad = eng._gram_ds_grad, None, None      # formed by gram_closing_backward
            elif blk.downsample is not None:
                add_bits = None
                gcd = eng.bn_backward(f"{name}.ds", g, "t5", g_bits=bits)
                dconv = blk.downsample[0]
                sub = dconv.stride[0] == 2 and dconv.kernel_size[0] == 1 and dconv.padding[0] == 0
                add = eng.conv_backward(f"{name}.ds", gcd, "t6", subgrid=sub)
                if sub:
                    add_hw = (add.shape[1], add.shape[2])
            # the gradient leaving this block is the output gradient of the previous block's closing stage: let the
            # epilogue that forms it also mask it and reduce it for that stage's BatchNorm backward
            g_stats = None
            if bi > 0:
                pname, pblk = blocks[bi - 1]
                plast = f"{pname}.{len(pblk.stages()) - 1}"
                if eng.can_fuse_residual_bn_backward(plast, f"{name}.0"):
                    g, g_stats = eng.conv_backward(f"{name}.0", gc, f"g{flip}", add=add, add_hw=add_hw, add_bits=add_bits,
                                                   fuse_bn=plast)
            if g_stats is None:
                g = eng.conv_backward(f"{name}.0", gc, f"g{flip}", add=add, add_hw=add_hw, add_bits=add_bits)
            flip ^= 1
            eng.end_block(bi)
            if on_done is not None and name.endswith(".0"):
                on_done(getattr(self, name.split(".")[0]))
        eng.begin_block(-1)
        if eng.saved["stem"]["pool_idx"] is not None:
            gc = eng.bn_pool_backward("stem", g, "t0")
        else:
            g = eng.maxpool_backward("pool", g, "mp")
            gc = eng.bn_backward("stem", g, "t0")
        eng.conv_backward("stem", gc, None)
        if on_done is not None:
            on_done(self.conv1)
            on_done(self.bn1)


_RESNETS = {
    "resnet18": (_Basic, (2, 2, 2, 2)),
    "resnet34": (_Basic, (3, 4, 6, 3)),
    "resnet50": (_Bottle, (3, 4, 6, 3)),
    "resnet101": (_Bottle, (3, 4, 23, 3)),
    "resnet152": (_Bottle, (3, 8, 36, 3)),
    # reduced members of the family used by the fast parity tests (same blocks, one per stage)
    "resnet_tiny_basic": (_Basic, (1, 1, 1, 1)),
    "resnet_tiny_bottleneck": (_Bottle, (1, 1, 1, 1)),
}


def create_backbone(name: str, pretrained: bool = False) -> nn.Module:
    """Counterpart of timm.create_model(name, pretrained=..., num_classes=0) for the families this engine runs."""
    key = name.lower()
    if pretrained:
        raise RuntimeError(f"pretrained=True needs timm's weight hub, which this offline engine does not ship; "
                           f"load a timm-format state_dict through cfg.model['checkpoint'] instead ({name})")
    if key in _RESNETS:
        blk, layers = _RESNETS[key]
        return HipResNet(blk, layers)
    from .vit import create_vit
    m = create_vit(key)
    if m is None:
        raise NotImplementedError(f"backbone {name!r} is not implemented by the HIP engine "
                                  f"(available: {sorted(_RESNETS)} + vit_{{small,base,large}}_patch16_224)")
    return m
