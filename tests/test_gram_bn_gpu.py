"""Gram form of the bottleneck closing stage (csrc/grambn.hip, nkb_conv_affine_residual, nkb_conv_dgrad_bn_cat) through the C ABI
against float64 torch on the SAME bf16 operands: the stage is timm Bottleneck conv3 -> bn3 -> += shortcut -> act3
(/root/reference/nkb_classification/model.py:82, engine.py:48,55-58).  The algebra itself is pinned on the CPU by
tests/test_gram_bn_math.py; here the kernels are held to it.

Tolerances: statistics 2e-4 relative (fp32 sums of exact bf16 products), stored bf16 outputs within bf16 rounding of the float64
value, weight / BatchNorm gradients 2e-3 of their norm (no bf16 rounding on that path), the data gradient 1e-2 of its norm (its
filter [k1 W ; Q] is rounded to bf16)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from nkb_classification import hip  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("N,H,w,proj,gamma_zero", [(8, 14, 64, False, False), (4, 28, 128, True, False), (16, 7, 256, False, False),
                                                   (32, 7, 512, True, False), (8, 14, 64, False, True)])
def test_gram_closing_stage_forward_and_backward(N, H, w, proj, gamma_zero):
    torch.manual_seed(w + N)
    d = hip.BF16
    M, co = N * H * H, 4 * w
    eps, mom = 1e-5, 0.1
    # stage before: a = relu(bn2(c2)) with scale / shift / mean of our choosing (only the mask and the sums depend on them)
    c2 = torch.randn(M, w, device=DEV).to(BF)
    sc2 = (torch.rand(w, device=DEV) + 0.5)
    sh2 = torch.randn(w, device=DEV) * 0.3
    mean2 = torch.randn(w, device=DEV) * 0.1
    a = torch.relu((c2.float() * sc2 + sh2).to(BF).float()).to(BF)
    W = (torch.randn(co, w, device=DEV) * (1.0 / w ** 0.5)).to(BF)
    gamma = torch.zeros(co, device=DEV) if gamma_zero else torch.randn(co, device=DEV) * 0.5 + 1.0
    beta = torch.randn(co, device=DEV) * 0.2
    res = torch.randn(M, co, device=DEV).to(BF)
    rsc = (torch.rand(co, device=DEV) + 0.5) if proj else None
    rsh = torch.randn(co, device=DEV) * 0.2 if proj else None
    rm, rv = torch.zeros(co, device=DEV), torch.ones(co, device=DEV)

    # ---- float64 reference of the forward pass -------------------------------------------------------------------------------
    a64, W64 = a.double(), W.double()
    c64 = a64 @ W64.t()
    mean64, var64 = c64.mean(0), c64.var(0, unbiased=False)
    r64 = 1.0 / torch.sqrt(var64 + eps)
    res_eff = res.double() if not proj else (res.float() * rsc + rsh).to(BF).double()
    o64 = torch.relu((c64 - mean64) * r64 * gamma.double() + beta.double() + res_eff)

    # ---- HIP forward: Gram matrix, statistics, the closing convolution ---------------------------------------------------------
    gs = torch.zeros(w * w + w, device=DEV)
    G, s = gs[:w * w], gs[w * w:]
    work = torch.empty(hip.conv_wgrad_workspace(d, N=N, P=H, Q=H, Cin=w, Cout=w, has_bias=True), device=DEV)
    hip.conv_wgrad(d, a, a, G, N=N, H=H, W=H, Cin=w, ldx=w, P=H, Q=H, Cout=w, lddy=w, dbias=s, workspace=work)
    assert _rel(G.view(w, w), a64.t() @ a64) < 1e-5 and _rel(s, a64.sum(0)) < 1e-5
    bnv = torch.empty(4, co, device=DEV)
    cov, mu, T = torch.empty(w * w, device=DEV), torch.empty(w, device=DEV), torch.empty(co, w, device=DEV)
    hip.gram_bn_stats(d, W, G, s, M, w, co, gamma, beta, rm, rv, mom, eps, cov, mu, T, bnv[0], bnv[1], bnv[2], bnv[3])
    assert _rel(bnv[2], mean64) < 2e-4
    assert _rel(bnv[3], r64) < 2e-4
    assert _rel(rm, mom * mean64) < 2e-4
    assert _rel(rv, 0.9 + mom * var64 * M / (M - 1)) < 2e-4
    assert _rel(bnv[0], gamma.double() * r64) < 2e-4 or gamma_zero
    y = torch.empty(M, co, device=DEV, dtype=BF)
    bits = torch.zeros(M, co // 8, device=DEV, dtype=torch.uint8)
    hip.conv_affine_residual(d, a, W, y, bnv[0], bnv[1], res, co, rsc, rsh, bits, N=N, H=H, W=H, Cin=w, ldx=w, P=H, Q=H, Cout=co, ldy=co)
    torch.cuda.synchronize()
    err = (y.double() - o64).abs()
    assert (err <= 1.5e-2 * o64.abs() + 1.5e-2).all(), err.max().item()
    unpacked = ((bits.view(M, co // 8, 1) >> torch.arange(8, device=DEV, dtype=torch.uint8)) & 1).view(M, co).bool()
    assert torch.equal(unpacked, y > 0)

    # ---- backward: g = masked output gradient (mask from the HIP output so both sides clamp the same elements) ------------------
    up = torch.randn(M, co, device=DEV).to(BF)
    g = (up.float() * (y > 0)).to(BF)
    g64 = g.double()
    dbeta64 = g64.sum(0)
    xhat = (c64 - mean64) * r64
    dgamma64 = (g64 * xhat).sum(0)
    dc64 = gamma.double() * r64 * (g64 - dbeta64 / M - xhat * dgamma64 / M)
    dW64 = dc64.t() @ a64
    mask2 = (c2.float() * sc2 + sh2).to(BF).float() > 0
    da64 = (dc64 @ W64) * mask2

    tiles = 1
    gstats = torch.zeros(hip.bn_stats_floats(tiles, co), device=DEV)
    gstats[:co] = g.float().sum(0)
    R = torch.zeros(co, w, device=DEV)
    work = torch.empty(hip.conv_wgrad_workspace(d, N=N, P=H, Q=H, Cin=w, Cout=co), device=DEV)
    hip.conv_wgrad(d, g, a, R, N=N, H=H, W=H, Cin=w, ldx=w, P=H, Q=H, Cout=co, lddy=co, workspace=work)
    dgamma, dbeta, dW = torch.zeros(co, device=DEV), torch.zeros(co, device=DEV), torch.zeros(co, w, device=DEV)
    wcat = torch.empty(w, co + w, device=DEV, dtype=BF)
    cbias, coef = torch.empty(w, device=DEV), torch.empty(hip.gram_bn_backward_ws(w, co), device=DEV)
    hip.gram_bn_backward(d, W, R, T, mu, gstats, tiles, M, w, co, gamma, bnv[2], bnv[3], dgamma, dbeta, dW, wcat, cbias, coef)
    torch.cuda.synchronize()
    scale_ref = max(dgamma64.norm().item(), 1e-6)
    assert (dgamma.double() - dgamma64).norm().item() < 2e-3 * scale_ref + 1e-3
    assert _rel(dbeta, dbeta64) < 1e-4
    assert torch.isfinite(dW).all() and torch.isfinite(wcat.float()).all()
    if gamma_zero:
        assert dW.abs().max().item() == 0.0       # gamma = 0: no gradient reaches the convolution
    else:
        assert _rel(dW, dW64) < 2e-3

    st_tiles = hip.stat_tiles(d, M, w)
    stats2 = torch.zeros(hip.bn_stats_floats(st_tiles, w), device=DEV)
    da = torch.empty(M, w, device=DEV, dtype=BF)
    hip.conv_dgrad_bn_cat(d, g, co, co, a, w, w, wcat, cbias, da, c2, sc2, sh2, mean2, stats2, M, w, w)
    torch.cuda.synchronize()
    if gamma_zero:
        assert da.float().abs().max().item() == 0.0
        return
    assert _rel(da, da64) < 1e-2
    # the fused epilogue's sums: sum g', sum g' (c2 - mean2) per channel of the stage before
    part = stats2[:st_tiles * 2 * w].view(st_tiles, 2, w).double().sum(0)
    daf = da.double()
    assert _rel(part[0], daf.sum(0)) < 1e-3
    assert _rel(part[1], (daf * (c2.double() - mean2.double())).sum(0)) < 1e-3

    # ---- the two-launch form: t = g . (k1 W) from the forward scale alone, then da = t + a . Q + cbias ----------------------------
    wk1 = torch.empty(w, co, device=DEV, dtype=BF)
    hip.gram_k1w(d, W, bnv[0], w, co, wk1)
    assert torch.equal(wk1, wcat[:, :co].contiguous())
    t = torch.empty(M, w, device=DEV, dtype=BF)
    hip.conv_gemm(d, 0, g, wk1, t, N=M, H=1, W=1, Cin=co, ldx=co, P=1, Q=1, Cout=w, ldy=w)
    q = torch.empty(w, w, device=DEV, dtype=BF)
    cbias2 = torch.empty(w, device=DEV)
    dg2, db2, dW2 = torch.zeros(co, device=DEV), torch.zeros(co, device=DEV), torch.zeros(co, w, device=DEV)
    hip.gram_bn_backward(d, W, R, T, mu, gstats, tiles, M, w, co, gamma, bnv[2], bnv[3], dg2, db2, dW2, None, cbias2, coef, q=q)
    torch.cuda.synchronize()
    assert torch.equal(dW2, dW) and torch.equal(cbias2, cbias) and torch.equal(q, wcat[:, co:].contiguous())
    stats3 = torch.zeros_like(stats2)
    da2 = torch.empty(M, w, device=DEV, dtype=BF)
    hip.conv_dgrad_bn_add(d, a, w, w, q, cbias2, t, w, da2, c2, sc2, sh2, mean2, stats3, M, w, w)
    torch.cuda.synchronize()
    assert _rel(da2, da64) < 1.2e-2
    part3 = stats3[:st_tiles * 2 * w].view(st_tiles, 2, w).double().sum(0)
    assert _rel(part3[0], da2.double().sum(0)) < 1e-3


@pytest.mark.parametrize("rows,C", [(8 * 56 * 56, 64), (4 * 28 * 28 + 37, 128), (100, 64), (5000, 128)])
def test_bn_apply_gram_matches_bn_apply_and_the_gram_matrix(rows, C):
    """nkb_bn_apply_gram: y bit-identical to nkb_bn_apply(relu), Gram matrix / column sums of the STORED y to fp32 summation error,
    bit-identical across repeats (ordered slab reduction)."""
    torch.manual_seed(rows + C)
    d = hip.BF16
    c = torch.randn(rows, C, device=DEV).to(BF)
    scale, shift = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3
    y_ref = torch.empty_like(c)
    hip.bn_apply(d, c, None, y_ref, scale, shift, rows, C, True)
    outs = []
    for _ in range(2):
        y = torch.empty_like(c)
        gs = torch.full((C * C + C,), float("nan"), device=DEV)
        work = torch.empty(hip.bn_apply_gram_ws(rows, C), device=DEV)
        hip.bn_apply_gram(d, c, y, scale, shift, rows, C, gs, work)
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int16), y_ref.view(torch.int16))
        outs.append(gs.clone())
    assert torch.equal(outs[0], outs[1])
    y64 = y_ref.double()
    assert _rel(outs[0][:C * C].view(C, C), y64.t() @ y64) < 1e-5
    assert _rel(outs[0][C * C:], y64.sum(0)) < 1e-5


def test_gram_form_in_the_model_matches_separate_passes(monkeypatch):
    """The same reduced bottleneck ResNet, same weights, same batch, bf16: train step with the Gram-form closing stages against
    the separate bn_apply / bn_backward passes — logits and every gradient agree to bf16 noise, and the Gram path really ran."""
    from nkb_classification import hipnet
    from nkb_classification.model import get_model
    monkeypatch.setattr(hipnet, "_GRAM_MAX_C", 512)        # all three eligible widths (64, 128: fused Gram pass; 256: nkb_conv_wgrad)

    def run(gram):
        cfg_model = dict(model="resnet_tiny_bottleneck", pretrained=False, backbone_dropout=0.0, classifier_dropout=0.0,
                         classifier_initialization="kaiming_normal_", task="single")
        torch.manual_seed(0)
        model = get_model(cfg_model, ["a", "b", "c"], DEV)
        gen = torch.Generator().manual_seed(5)
        with torch.no_grad():                   # zero_init_last would silence the very branches under test
            for p in model.parameters():
                if p.dim() == 1:
                    p.copy_((torch.rand(p.shape, generator=gen) * 0.5 + 0.5).to(p.device))
        model.train()
        eng = model._engine(torch.device(DEV), torch.bfloat16)
        eng.gram_bn = gram
        x = torch.randn(16, 3, 64, 64, generator=gen).to(DEV)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x)
            loss = torch.nn.functional.cross_entropy(out.float(), torch.arange(16, device=DEV) % 3)
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}
        used = sum(1 for sv in eng.saved.values() if isinstance(sv, dict) and sv.get("gram") is not None)
        return out.detach().float().clone(), grads, used

    o1, g1, used1 = run(True)
    o0, g0, used0 = run(False)
    assert used1 == 3 and used0 == 0
    assert _rel(o1, o0) < 3e-2
    flat1 = torch.cat([g1[k].flatten() for k in sorted(g1)])
    flat0 = torch.cat([g0[k].flatten() for k in sorted(g0)])
    cos = torch.nn.functional.cosine_similarity(flat1, flat0, dim=0).item()
    assert cos > 0.995, cos
