"""Row-balanced, DMA-pipelined 3x3 core (csrc/convp.hip: nkb_convp_fwd / nkb_convp_dgrad_bn) through the C ABI: against torch's CPU
conv2d (forward, data gradient), against the 128 x 128 implicit-GEMM kernel it replaces in the train step (same operands: the fp32
accumulation order is the same, so the bf16 outputs are bit-identical), and against the unfused BatchNorm backward; repeat launches
are bit-identical (no atomics); shapes outside its eligibility are refused.  timm Bottleneck / BasicBlock conv2 reached from
/root/reference/nkb_classification/engine.py:48, 55-58 via model.py:82."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from nkb_classification import hip  # noqa: E402

DEV = "cuda"
T = torch.bfloat16
D = hip.BF16

# N, H, W, Cin, Cout: 128- and 256-channel tiles, sub-tiles with 4 ... 16 fragments, a pixel count that is no multiple of 16,
# several sub-tiles per workgroup (the last case: 640 rows per workgroup on a 256-CU chip)
# the 64 -> 64 channel form (filter resident in registers / LDS): one partial sub-tile, ragged rows, 3.25 sub-tiles per workgroup
SHAPES = [(6, 28, 28, 64, 128), (24, 14, 14, 128, 256), (9, 23, 23, 64, 128), (90, 7, 7, 128, 512), (40, 28, 28, 64, 256),
          (209, 28, 28, 64, 128), (256, 14, 14, 64, 256), (4, 56, 56, 64, 64), (5, 37, 29, 64, 64), (67, 56, 56, 64, 64)]


@pytest.fixture(autouse=True)
def _narrow_tiles_too():
    """The 128-channel form and the 64-channel data gradient are not taken by the train step (nkb_convp_config): these tests cover
    them as well."""
    hip.convp_config(True, True, True, True)
    yield
    hip.convp_config(True, False)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _geom(N, H, W, Cin, Cout):
    return dict(N=N, H=H, W=W, Cin=Cin, ldx=Cin, Cout=Cout, ldy=Cout)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_convp_forward_matches_torch_and_the_128x128_kernel(shape):
    N, H, W, Cin, Cout = shape
    torch.manual_seed(sum(shape))
    x = torch.randn(N, Cin, H, W).to(T).float()
    w = (torch.randn(Cout, Cin, 3, 3) / math.sqrt(9 * Cin)).to(T).float()
    g = _geom(N, H, W, Cin, Cout)
    tiles = hip.convp_tiles(D, 0, R=3, S=3, stride=1, pad=1, **g)
    assert 0 < tiles <= 256
    xd, wd = _nhwc(x).to(DEV, T), _nhwc(w).to(DEV, T)
    y = torch.full((N, H, W, Cout), float("nan"), device=DEV, dtype=T)
    stats = torch.full((hip.bn_stats_floats(tiles, Cout),), float("nan"), device=DEV)
    n0 = hip.kernel_launches("convp")
    hip.convp_fwd(D, xd, wd, y, stats, tiles=tiles, **g)
    assert hip.kernel_launches("convp") == n0 + 1
    # the kernel it replaces, same operands
    y0 = torch.empty_like(y)
    t0 = hip.stat_tiles(D, N * H * W, Cout)
    s0 = torch.zeros(hip.bn_stats_floats(t0, Cout), device=DEV)
    hip.conv_gemm(D, 0, xd, wd, y0, stats=s0, N=N, H=H, W=W, Cin=Cin, ldx=Cin, P=H, Q=W, Cout=Cout, ldy=Cout, R=3, S=3, stride=1, pad=1)
    y2, stats2 = torch.empty_like(y), torch.empty_like(stats)
    hip.convp_fwd(D, xd, wd, y2, stats2, tiles=tiles, **g)
    torch.cuda.synchronize()
    ref = _nhwc(F.conv2d(x, w, padding=1))
    torch.testing.assert_close(y.float().cpu(), ref, rtol=2e-2, atol=2e-2 * math.sqrt(9 * Cin) / 4)
    assert torch.equal(y, y0)                                   # same k order, fp32 accumulators: bit-identical to conv_igemm
    assert torch.equal(y, y2) and torch.equal(stats[: tiles * 2 * Cout], stats2[: tiles * 2 * Cout])      # repeat launches: bit-identical
    st = stats[: tiles * 2 * Cout].view(tiles, 2, Cout).double().sum(0).cpu()
    yf = y.float().cpu().double().reshape(-1, Cout)
    torch.testing.assert_close(st[0], yf.sum(0), rtol=1e-5, atol=1e-3 * math.sqrt(N * H * W))
    torch.testing.assert_close(st[1], (yf * yf).sum(0), rtol=1e-5, atol=1e-3 * math.sqrt(N * H * W))


@pytest.mark.parametrize("shape", SHAPES[:5], ids=lambda s: "x".join(map(str, s)))
def test_convp_dgrad_with_fused_bn_backward_matches_torch_and_unfused(shape):
    """nkb_convp_dgrad_bn + nkb_bn_backward_from_stats == torch's conv2d data gradient through relu(bn(c)), and == the unfused pair
    nkb_conv_gemm(mode 1) + nkb_bn_backward on the same operands."""
    N, H, W, C, Co = shape                        # the consumer conv maps C (the BatchNorm stage's channels) -> Co; its data gradient maps Co -> C
    if C % 128:
        C, Co = Co, C                             # the produced gradient has C channels: they are the kernel's output channels
    torch.manual_seed(sum(shape) + 1)
    rows = N * H * W
    c = torch.randn(N, H, W, C, device=DEV).to(T)
    dy = torch.randn(N, H, W, Co, device=DEV).to(T)
    wt = (torch.randn(C, 3, 3, Co, device=DEV) / math.sqrt(9 * Co)).to(T)      # data-gradient layout [Cin][R][S][Cout]
    gamma = torch.rand(C, device=DEV) + 0.5
    cf = c.float().reshape(rows, C)
    mean = cf.mean(0).contiguous()
    invstd = (cf.var(0, unbiased=False) + 1e-5).rsqrt().contiguous()
    scale = (gamma * invstd).contiguous()
    shift = (0.05 - mean * scale).contiguous()
    g = dict(N=N, H=H, W=W, Cin=Co, ldx=Co, Cout=C, ldy=C)
    tiles = hip.convp_tiles(D, 1, R=3, S=3, stride=1, pad=1, **g)
    assert tiles > 0
    # unfused pair
    geom = dict(N=N, H=H, W=W, Cin=Co, ldx=Co, P=H, Q=W, Cout=C, ldy=C, R=3, S=3, stride=1, pad=1)
    g0 = torch.empty(N, H, W, C, device=DEV, dtype=T)
    hip.conv_gemm(D, 1, dy, wt, g0, **geom)
    graw = g0.clone()
    dg0, db0 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dc0 = torch.empty_like(c)
    work = torch.empty(hip.bn_backward_ws(rows, C), device=DEV)
    hip.bn_backward(D, g0, c, None, mean, invstd, gamma, rows, C, dg0, db0, dc0, g0, work, fscale=scale, fshift=shift)
    # fused, new core
    g1 = torch.full_like(g0, float("nan"))
    stats = torch.full((hip.bn_stats_floats(tiles, C),), float("nan"), device=DEV)
    hip.convp_dgrad_bn(D, dy, wt, g1, c, scale, shift, mean, stats, tiles=tiles, **g)
    dg1, db1 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dc1 = torch.empty_like(c)
    sums = torch.empty(2 * C, device=DEV)
    hip.bn_backward_from_stats(D, g1, c, stats, tiles, mean, invstd, gamma, rows, C, dg1, db1, dc1, sums)
    torch.cuda.synchronize()
    assert torch.equal(g0, g1)                    # masked gradient (bn_backward wrote its mask back into g0)
    torch.testing.assert_close(db1, db0, rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(dg1, dg0, rtol=1e-3, atol=2e-2)
    torch.testing.assert_close(dc1.float(), dc0.float(), rtol=2e-2, atol=1e-2)
    # torch: gradient of conv2d w.r.t. its input (the unmasked gradient), then the mask relu(bn(c)) > 0
    xin = torch.zeros(N, C, H, W, requires_grad=True)
    wt_cpu = wt.float().cpu().permute(3, 0, 1, 2).contiguous()                # [Cout = Co][Cin = C][3][3]
    F.conv2d(xin, wt_cpu, padding=1).backward(dy.float().cpu().permute(0, 3, 1, 2).contiguous())
    torch.testing.assert_close(graw.float().cpu(), _nhwc(xin.grad), rtol=2e-2, atol=2e-2 * math.sqrt(9 * Co) / 4)


def test_convp_refuses_what_it_cannot_run():
    ok = dict(N=8, H=28, W=28, Cin=64, ldx=64, Cout=128, ldy=128, R=3, S=3, stride=1, pad=1)
    assert hip.convp_tiles(D, 0, **ok) > 0
    for bad in (dict(R=1, S=1, pad=0), dict(stride=2), dict(Cin=32, ldx=32), dict(Cout=64, ldy=64, Cin=128, ldx=128), dict(N=1), dict(Cout=192, ldy=192)):
        assert hip.convp_tiles(D, 0, **{**ok, **bad}) == 0, bad
    hip.convp_config(True, False)                 # the train step's envelope: 256-channel tiles only
    assert hip.convp_tiles(D, 0, **ok) == 0 and hip.convp_tiles(D, 0, **{**ok, "Cout": 256, "ldy": 256}) > 0
    hip.convp_config(False, False)
    assert hip.convp_tiles(D, 0, **{**ok, "Cout": 256, "ldy": 256}) == 0
    c64 = dict(ok, H=56, W=56, Cout=64, ldy=64)
    hip.convp_config(True, False)                 # ... and the 64-channel form forward only
    assert hip.convp_tiles(D, 0, **c64) > 0 and hip.convp_tiles(D, 1, **c64) == 0
    hip.convp_config(True, False, False)
    assert hip.convp_tiles(D, 0, **c64) == 0
    hip.convp_config(True, True, True, True)
    assert hip.convp_tiles(D, 1, **c64) > 0
    assert hip.convp_tiles(hip.F32, 0, **ok) == 0
    x = torch.zeros(1, 28, 28, 64, device=DEV, dtype=T)
    with pytest.raises(RuntimeError, match="not eligible"):
        hip.convp_fwd(D, x, x, x, torch.zeros(8, device=DEV), N=1, H=28, W=28, Cin=64, ldx=64, Cout=128, ldy=128, tiles=1)


def test_backward_grids_leave_reserved_cus_to_a_collective():
    """nkb_rowres_reserve_cus (what a multi-rank GradReducer calls before the first step): the data-gradient kernel and the streamed
    g^T a size their grids for #CUs - 32 — fewer partial rows / slabs, the same numbers."""
    shape = (256, 14, 14, 256, 256)
    N, H, W, C, Co = shape
    g = dict(N=N, H=H, W=W, Cin=Co, ldx=Co, Cout=C, ldy=C)
    t0 = hip.convp_tiles(D, 1, R=3, S=3, stride=1, pad=1, **g)
    f0 = hip.convp_tiles(D, 0, R=3, S=3, stride=1, pad=1, **g)
    w0 = hip.gramr_workspace(D, 802816, 256, 64)
    torch.manual_seed(3)
    c = torch.randn(N, H, W, C, device=DEV).to(T)
    dy = torch.randn(N, H, W, Co, device=DEV).to(T)
    wt = (torch.randn(C, 3, 3, Co, device=DEV) / math.sqrt(9 * Co)).to(T)
    scale, shift, mean = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1, torch.randn(C, device=DEV) * 0.1
    out = []
    try:
        for reserve in (0, 32):
            hip.rowres_reserve_cus(reserve)
            tiles = hip.convp_tiles(D, 1, R=3, S=3, stride=1, pad=1, **g)
            g1 = torch.full((N, H, W, C), float("nan"), device=DEV, dtype=T)
            stats = torch.full((hip.bn_stats_floats(tiles, C),), float("nan"), device=DEV)
            hip.convp_dgrad_bn(D, dy, wt, g1, c, scale, shift, mean, stats, tiles=tiles, **g)
            out.append((tiles, g1, stats[: tiles * 2 * C].view(tiles, 2, C).double().sum(0)))
            if reserve:                                                 # a buffer (or a recorded plan) sized under the other setting is refused
                with pytest.raises(RuntimeError, match="partial-sum rows"):
                    hip.convp_dgrad_bn(D, dy, wt, g1, c, scale, shift, mean, stats, tiles=t0, **g)
        assert out[0][0] == t0 and out[1][0] < t0                       # 242 -> 224 workgroups on a 256-CU chip
        assert hip.convp_tiles(D, 0, R=3, S=3, stride=1, pad=1, **g) == f0  # the forward grid is not touched
        assert 0 < hip.gramr_workspace(D, 802816, 256, 64) < w0
        assert torch.equal(out[0][1], out[1][1])
        torch.testing.assert_close(out[0][2], out[1][2], rtol=1e-5, atol=1e-2)
    finally:
        hip.rowres_reserve_cus(0)
