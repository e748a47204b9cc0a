"""CPU check (float64) of the algebra behind the Gram form of a bottleneck's closing stage ("Gram-BN"):

    c = y2 @ W^T,  o = relu(bn(c) + res)       (timm Bottleneck conv3 -> bn3 -> += shortcut -> act3;
                                                 /root/reference/nkb_classification/model.py:82 reached from engine.py:48,55-58)

With G = y2^T y2 and s = colsum(y2) the batch statistics of c follow WITHOUT c (mean = W mu, var = diag(W Cov W^T)), and with
R = g^T y2 the whole BatchNorm + conv backward follows without c or dc:

    sum_p g c = rowdot(W, R);   dW = k1 R + M k2 T - gamma r dbeta mu^T   (T = W Cov);   dy2 = g (k1 W) + y2 Q + const

The HIP path (csrc/grambn.hip + the K-concatenated data gradient) implements exactly these formulas; this test pins them
against torch autograd so a GPU mismatch can only be an implementation error, not a derivation error."""
import torch


def _case(M, w, co, seed, gamma_zero=False):
    g = torch.Generator().manual_seed(seed)
    y2 = torch.relu(torch.randn(M, w, generator=g, dtype=torch.float64) + 0.3)
    W = torch.randn(co, w, generator=g, dtype=torch.float64) * 0.2
    gamma = torch.randn(co, generator=g, dtype=torch.float64)
    if gamma_zero:
        gamma.zero_()
    beta = torch.randn(co, generator=g, dtype=torch.float64) * 0.1
    res = torch.randn(M, co, generator=g, dtype=torch.float64)
    up = torch.randn(M, co, generator=g, dtype=torch.float64)       # gradient arriving at the block output
    return y2, W, gamma, beta, res, up


def _autograd(y2, W, gamma, beta, res, up, eps):
    y2 = y2.clone().requires_grad_(True)
    W = W.clone().requires_grad_(True)
    gamma = gamma.clone().requires_grad_(True)
    beta = beta.clone().requires_grad_(True)
    c = y2 @ W.t()
    mean, var = c.mean(0), c.var(0, unbiased=False)
    o = torch.relu((c - mean) / torch.sqrt(var + eps) * gamma + beta + res)
    (o * up).sum().backward()
    return o.detach(), mean.detach(), var.detach(), y2.grad, W.grad, gamma.grad, beta.grad


def _gram_form(y2, W, gamma, beta, res, up, eps):
    M = y2.shape[0]
    # forward: statistics from the Gram matrix
    G, s = y2.t() @ y2, y2.sum(0)
    mu = s / M
    Cov = G / M - torch.outer(mu, mu)
    T = W @ Cov
    mean = W @ mu
    var = (T * W).sum(1)
    r = 1.0 / torch.sqrt(var + eps)
    scale, shift = gamma * r, beta - mean * gamma * r
    o = torch.relu((y2 @ W.t()) * scale + shift + res)          # epilogue of the closing convolution: c is never stored
    # backward: g = masked gradient of the pre-activation sum
    g = up * (o > 0)
    dbeta = g.sum(0)
    R = g.t() @ y2                                               # the "raw" weight gradient
    gc = (W * R).sum(1)                                          # sum_p g c
    dgamma = r * (gc - mean * dbeta)
    k1 = gamma * r
    k2 = -gamma * r * r * dgamma / M
    k3 = -gamma * r * dbeta / M - k2 * mean
    dW = k1[:, None] * R + M * k2[:, None] * T - (gamma * r * dbeta)[:, None] * mu[None, :]
    Q = W.t() @ (k2[:, None] * W)
    const = k3 @ W
    dy2 = torch.cat([g, y2], 1) @ torch.cat([k1[:, None] * W, Q], 0) + const       # the K-concatenated data gradient
    return o, mean, var, dy2, dW, dgamma, dbeta


def test_gram_form_matches_autograd():
    eps = 1e-5
    for (M, w, co, seed) in [(512, 16, 64, 0), (300, 8, 32, 1), (1024, 32, 128, 2)]:
        case = _case(M, w, co, seed)
        ref = _autograd(*case, eps)
        got = _gram_form(*case, eps)
        for name, a, b in zip(("o", "mean", "var", "dy2", "dW", "dgamma", "dbeta"), ref, got):
            assert torch.allclose(a, b, rtol=1e-9, atol=1e-9), (name, (a - b).abs().max().item())


def test_gram_form_with_zero_gamma():
    """timm zero-initialises the closing BatchNorm's weight (zero_init_last): every coefficient must stay finite."""
    case = _case(256, 8, 32, 3, gamma_zero=True)
    ref = _autograd(*case, 1e-5)
    got = _gram_form(*case, 1e-5)
    for a, b in zip(ref, got):
        assert torch.isfinite(b).all()
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-9)
