"""The pixel-resident 1x1 expansion kernel (csrc/conv1p.hip), the ring-buffered stem convolution and weight gradient (csrc/stemp.hip) and the
streamed g^T a product (csrc/gramr.hip) through the C ABI:
both against torch's fp32 convolution of the same bf16 operands (the oracle's arithmetic for these layers:
/root/reference/nkb_classification/model.py:82 builds them, engine.py:48 runs them), against the tile kernels they replace in the train
step — same MFMA instruction, same summation order: bit-identical outputs — and their BatchNorm partial sums against sums of the stored
outputs.  Ragged pixel counts, workgroups with fewer rows than the rest, image sizes that are no multiple of anything."""
import math

import pytest
import torch
import torch.nn.functional as F

from nkb_classification import hip

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
T = torch.bfloat16
D = hip.BF16


def _stats_of(y, tiles, co, stats):
    yf = y.float().reshape(-1, co)
    s = stats[: tiles * 2 * co].view(tiles, 2, co).double().sum(0).cpu()
    torch.testing.assert_close(s[0], yf.double().sum(0).cpu(), rtol=1e-4, atol=1e-2)
    torch.testing.assert_close(s[1], (yf.double() ** 2).sum(0).cpu(), rtol=1e-4, atol=1e-2)


# N, H (square maps), Cin, Cout: layer3-like tiles of 3 .. 13 fragments, a last workgroup with fewer rows
# ... and the streaming form of the reduction stage (Cin >= 512 -> Cout <= Cin / 2): one and two channel blocks, ragged rows
C1_SHAPES = [(60, 14, 256, 512), (131, 14, 256, 1024), (256, 14, 256, 1024), (61, 15, 256, 768), (75, 13, 256, 512),
             (256, 14, 1024, 256), (131, 14, 1024, 256), (203, 15, 1024, 512), (256, 14, 512, 256)]


@pytest.mark.parametrize("shape", C1_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_conv1p_matches_torch_and_the_tile_kernel(shape):
    N, H, ci, co = shape
    M = N * H * H
    tiles = hip.conv1p_tiles(D, M, ci, ci, co, co)
    assert 0 < tiles <= 256, shape
    torch.manual_seed(sum(shape))
    x = torch.randn(N, H, H, ci).to(T)
    w = (torch.randn(co, 1, 1, ci) / math.sqrt(ci)).to(T)
    xd, wd = x.to(DEV), w.to(DEV)
    y = torch.full((N, H, H, co), float("nan"), device=DEV, dtype=T)
    stats = torch.full((hip.bn_stats_floats(tiles, co),), float("nan"), device=DEV)
    n0 = hip.kernel_launches("conv1p")
    hip.conv1p_fwd(D, xd, wd, y, stats, M=M, Cin=ci, ldx=ci, Cout=co, ldy=co)
    assert hip.kernel_launches("conv1p") == n0 + 1
    # the kernel it replaces, same operands
    y0 = torch.empty_like(y)
    t0 = hip.stat_tiles(D, M, co)
    s0 = torch.zeros(hip.bn_stats_floats(t0, co), device=DEV)
    hip.conv_gemm(D, 0, xd, wd, y0, stats=s0, N=N, H=H, W=H, Cin=ci, ldx=ci, P=H, Q=H, Cout=co, ldy=co, R=1, S=1, stride=1, pad=0)
    assert torch.equal(y, y0)
    # torch, fp32 accumulation of the same bf16 operands (a sample of rows: the full product is 60 GFLOP on the host at bench size)
    rows = torch.randperm(M)[:4096]
    ref = x.reshape(M, ci)[rows].float() @ w.reshape(co, ci).float().t()
    torch.testing.assert_close(y.reshape(M, co)[rows.to(DEV)].float().cpu(), ref, rtol=2e-2, atol=2e-2)
    _stats_of(y, tiles, co, stats)
    # deterministic
    y2 = torch.empty_like(y)
    s2 = torch.empty_like(stats)
    hip.conv1p_fwd(D, xd, wd, y2, s2, M=M, Cin=ci, ldx=ci, Cout=co, ldy=co)
    assert torch.equal(y, y2) and torch.equal(stats[: tiles * 2 * co], s2[: tiles * 2 * co])


def test_conv1p_refuses_what_it_cannot_run():
    ok = dict(M=50176, Cin=256, ldx=256, Cout=1024, ldy=1024)
    assert hip.conv1p_tiles(D, **ok) > 0
    for bad in (dict(Cin=128, ldx=128), dict(Cout=384, ldy=384), dict(Cout=256, ldy=256), dict(M=1000), dict(Cin=768, ldx=768, Cout=512, ldy=512), dict(M=12544, Cin=512, ldx=512, Cout=2048, ldy=2048),
                dict(M=802816)):
        assert hip.conv1p_tiles(D, **{**ok, **bad}) == 0, bad
    assert hip.conv1p_tiles(hip.F32, **ok) == 0
    hip.convp_config(True, conv1p=False)
    try:
        assert hip.conv1p_tiles(D, **ok) == 0
        x = torch.zeros(50176, 256, device=DEV, dtype=T)
        with pytest.raises(RuntimeError, match="not eligible"):
            hip.conv1p_fwd(D, x, x, x, torch.zeros(8, device=DEV), **ok)
    finally:
        hip.convp_config(True)
    assert hip.conv1p_tiles(D, **ok) > 0


# N, H, W: the bench geometry, bands of output rows (small batches), odd sizes (a padded pixel column, ragged last fragment / band)
ST_SHAPES = [(2, 224, 224), (3, 64, 64), (5, 97, 131), (1, 224, 224), (7, 32, 250), (300, 224, 224)]


@pytest.mark.parametrize("shape", ST_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_stemp_matches_torch_and_the_tile_kernel(shape):
    N, H, W = shape
    co = 64
    tiles = hip.stemp_tiles(D, N, H, W, co)
    assert tiles > 0, shape
    torch.manual_seed(sum(shape))
    img = torch.randn(N, 3, H, W)
    w = torch.randn(co, 3, 7, 7) / math.sqrt(147)
    Wp = (W + 1) & ~1
    xp = torch.empty(N, H, Wp, 4, device=DEV, dtype=T)
    hip.stem_pack(D, img.to(DEV), xp, N, 3, H, W)
    wp = torch.empty(co, hip.stem_weight_cols(D), device=DEV, dtype=T)
    hip.stem_wprep(D, w.permute(0, 2, 3, 1).contiguous().to(DEV), wp, co, 3)
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.full((N, P, Q, co), float("nan"), device=DEV, dtype=T)
    stats = torch.full((hip.bn_stats_floats(tiles, co),), float("nan"), device=DEV)
    n0 = hip.kernel_launches("stemp")
    hip.stemp_conv(D, xp, wp, y, stats, N, H, W, co, co)
    assert hip.kernel_launches("stemp") == n0 + 1
    y0 = torch.empty_like(y)
    t0 = hip.stat_tiles(D, N * P * Q, co)
    s0 = torch.zeros(hip.bn_stats_floats(t0, co), device=DEV)
    hip.stem_conv(D, xp, wp, y0, s0, N, H, W, co, co)
    assert torch.equal(y, y0)
    if N <= 8:
        ref = F.conv2d(img.to(T).float(), w.to(T).float(), stride=2, padding=3).permute(0, 2, 3, 1)
        torch.testing.assert_close(y.float().cpu(), ref, rtol=2e-2, atol=2e-2)
    _stats_of(y, tiles, co, stats)
    y2 = torch.empty_like(y)
    s2 = torch.empty_like(stats)
    hip.stemp_conv(D, xp, wp, y2, s2, N, H, W, co, co)
    assert torch.equal(y, y2) and torch.equal(stats[: tiles * 2 * co], s2[: tiles * 2 * co])


@pytest.mark.parametrize("shape", ST_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_stemp_weight_gradient_matches_torch_and_the_generic_kernel(shape):
    """dW of the stem on the LDS ring (transposing reads of dY rows and of the overlapping windows): against torch's fp32 weight gradient
    of the same bf16 operands (small batches), against the generic split-over-pixels kernel (fp32 summation order differs), and twice
    (bit-identical: fixed order, no atomics)."""
    N, H, W = shape
    co = 64
    need = hip.stemp_wgrad_workspace(D, N, H, W, co)
    if shape == (7, 32, 250):
        assert need > 0                           # (157 KB of LDS: the widest rows that still fit)
    assert need > 0, shape
    torch.manual_seed(sum(shape) + 1)
    img = torch.randn(N, 3, H, W)
    Wp = (W + 1) & ~1
    xp = torch.empty(N, H, Wp, 4, device=DEV, dtype=T)
    hip.stem_pack(D, img.to(DEV), xp, N, 3, H, W)
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dy = torch.randn(N, P, Q, co).to(T)
    dyd = dy.to(DEV)
    cols = 224
    dw1 = torch.zeros(co, cols, device=DEV)
    work1 = torch.empty(need, device=DEV)
    n0 = hip.kernel_launches("stemp")
    hip.stemp_wgrad(D, dyd, xp, dw1, N, H, W, co, co, work1)
    assert hip.kernel_launches("stemp") == n0 + 1
    dw0 = torch.zeros(co, cols, device=DEV)
    work0 = torch.empty(hip.stem_wgrad_workspace(D, N, H, W, co), device=DEV)
    hip.stem_wgrad(D, dyd, xp, dw0, N, H, W, co, co, workspace=work0)
    scale = dw0.abs().max().item()
    torch.testing.assert_close(dw1, dw0, rtol=1e-4, atol=2e-5 * scale)
    if N <= 8:
        wt = torch.zeros(co, 3, 7, 7, requires_grad=True)
        F.conv2d(img.to(T).float(), wt, stride=2, padding=3).backward(dy.float().permute(0, 3, 1, 2).contiguous())
        folded = torch.zeros(co, 7, 7, 3, device=DEV)
        hip.stem_wfold(D, dw1, folded, co, 3)
        torch.testing.assert_close(folded.cpu(), wt.grad.permute(0, 2, 3, 1), rtol=1e-3, atol=1e-3 * scale)
    dw2 = torch.zeros(co, cols, device=DEV)
    hip.stemp_wgrad(D, dyd, xp, dw2, N, H, W, co, co, work1)
    assert torch.equal(dw1, dw2)


def test_stemp_refuses_what_it_cannot_run():
    assert hip.stemp_tiles(D, 8, 224, 224, 64) > 0
    assert hip.stemp_tiles(D, 8, 224, 224, 32) == 0 and hip.stemp_tiles(D, 8, 224, 300, 64) == 0 and hip.stemp_tiles(hip.F32, 8, 224, 224, 64) == 0
    assert hip.stemp_wgrad_workspace(D, 8, 224, 224, 64) == hip.stemp_tiles(D, 8, 224, 224, 64) * 64 * 224
    assert hip.stemp_wgrad_workspace(D, 8, 224, 224, 32) == 0
    hip.convp_config(True, stemp=False)
    try:
        assert hip.stemp_tiles(D, 8, 224, 224, 64) == 0
    finally:
        hip.convp_config(True)


# N, H (square maps), co, ci: both shapes of the Gram-form stages, ragged pixel counts (a last stage that is mostly zero page)
GR_SHAPES = [(8, 56, 256, 64), (7, 53, 256, 64), (32, 28, 512, 128), (33, 27, 512, 128), (64, 56, 256, 64)]


@pytest.mark.parametrize("shape", GR_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_gramr_matches_torch_and_the_generic_weight_gradient(shape):
    """R = g^T a streamed through LDS with both operands through the transposing read: against torch's fp32 product of the same bf16
    operands, against nkb_conv_wgrad_assign (different fp32 summation order), in its transposed + accumulating form, and twice."""
    N, H, co, ci = shape
    M = N * H * H
    need = hip.gramr_workspace(D, M, co, ci)
    assert need > 0, shape
    torch.manual_seed(sum(shape))
    g = torch.randn(M, co).to(T)
    a = torch.randn(M, ci).to(T)
    gd, ad = g.to(DEV), a.to(DEV)
    R1 = torch.full((co, ci), float("nan"), device=DEV)
    work = torch.empty(need, device=DEV)
    n0 = hip.kernel_launches("gramr")
    hip.gramr(D, gd, co, ad, ci, R1, M, co, ci, work)
    assert hip.kernel_launches("gramr") == n0 + 1
    ref = g.float().t() @ a.float()
    scale = ref.abs().max().item()
    torch.testing.assert_close(R1.cpu(), ref, rtol=1e-4, atol=2e-5 * scale)
    R0 = torch.full((co, ci), float("nan"), device=DEV)
    w0 = torch.empty(hip.conv_wgrad_workspace(D, N=N, P=H, Q=H, Cin=ci, Cout=co), device=DEV)
    hip.conv_wgrad(D, gd, ad, R0, N=N, H=H, W=H, Cin=ci, ldx=ci, P=H, Q=H, Cout=co, lddy=co, workspace=w0, assign=True)
    torch.testing.assert_close(R1, R0, rtol=1e-4, atol=2e-5 * scale)
    Dt = torch.ones(ci, co, device=DEV)
    hip.gramr(D, gd, co, ad, ci, Dt, M, co, ci, work, assign=False, transposed=True)
    torch.testing.assert_close(Dt.cpu(), 1.0 + ref.t(), rtol=1e-4, atol=2e-5 * scale)
    R2 = torch.empty_like(R1)
    hip.gramr(D, gd, co, ad, ci, R2, M, co, ci, work)
    assert torch.equal(R1, R2)


def test_gramr_refuses_what_it_cannot_run():
    assert hip.gramr_workspace(D, 802816, 256, 64) > 0 and hip.gramr_workspace(D, 200704, 512, 128) > 0
    for bad in ((802816, 256, 128), (802816, 128, 64), (8000, 256, 64), (802816, 64, 256)):
        assert hip.gramr_workspace(D, *bad) == 0, bad
    assert hip.gramr_workspace(hip.F32, 802816, 256, 64) == 0
    hip.convp_config(True, gramr=False)
    try:
        assert hip.gramr_workspace(D, 802816, 256, 64) == 0
    finally:
        hip.convp_config(True)
