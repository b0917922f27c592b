"""Torch restatement of the DKT hot path -- TEST INFRASTRUCTURE ONLY (PARITY UNPINNED, see dkt_oracle.py).

Two uses:
  * float64 + autograd: independent check of the closed-form gradients in dkt_oracle.py and of
    the HIP backward kernels (through F.normalize and bn_out as the reference's autograd does,
    methods/DKT.py:141-163).
  * float32, written in the STRUCTURE GPyTorch uses (per-class Python loop, dense evaluate ->
    torch.linalg.cholesky -> cholesky_solve, autograd backward): the `cpu_baseline` "port" that
    bench.py times on the GPU box's host cores (denominator A).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)


def one_vs_rest_targets(n_way, per_class, dtype=torch.float64):
    n = n_way * per_class
    y = -torch.ones(n_way, n, dtype=dtype)
    for c in range(n_way):
        y[c, c * per_class:(c + 1) * per_class] = 1.0
    return y


def base_matrix(za, zb, kernel, lengthscale=None):
    if kernel in ("linear", "cossim", "bncossim"):
        return za @ (za if zb is None else zb).T
    if kernel in ("rbf", "RBF"):
        zb_ = za if zb is None else zb
        # gpytorch Kernel.covar_dist: centre by x1's mean, norm expansion, clamp
        adj = za.mean(0, keepdim=True)
        a = (za - adj) / lengthscale
        b = (zb_ - adj) / lengthscale
        d2 = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T
        return torch.exp(-0.5 * d2.clamp_min(0.0))
    if kernel == "matern":       # MaternKernel(nu=2.5): sqrt(clamp(d2, 1e-30)) as gpytorch does
        zb_ = za if zb is None else zb
        adj = za.mean(0, keepdim=True)
        a = (za - adj) / lengthscale
        b = (zb_ - adj) / lengthscale
        d2 = (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * a @ b.T
        r = math.sqrt(5.0) * torch.sqrt(d2.clamp_min(1e-30))
        return (1.0 + r + r * r / 3.0) * torch.exp(-r)
    if kernel in ("poli1", "poli2"):     # PolynomialKernel: the offset rides in the `lengthscale` argument
        g = za @ (za if zb is None else zb).T
        return (g + lengthscale) ** (1 if kernel == "poli1" else 2)
    raise ValueError(kernel)


def spectral_mixture(za, zb, weights, means, scales):
    """SpectralMixtureKernel (DKT_regression.py:121-122) with torch ops, differentiable in every argument."""
    zb_ = za if zb is None else zb
    q = weights.numel()
    tau = za[:, None, :] - zb_[None, :, :]
    mu, sg = means.reshape(q, 1, 1, -1), scales.reshape(q, 1, 1, -1)
    res = torch.exp(-2.0 * math.pi ** 2 * (tau * sg) ** 2) * torch.cos(2.0 * math.pi * tau * mu)
    return (res.prod(-1) * weights.reshape(q, 1, 1)).sum(0)


def psd_safe_cholesky(k, jitter0=None, max_tries=3):
    l, info = torch.linalg.cholesky_ex(k)
    if int(info) == 0:
        return l
    if jitter0 is None:
        jitter0 = 1e-6 if k.dtype == torch.float32 else 1e-8
    eye = torch.eye(k.shape[-1], dtype=k.dtype)
    for i in range(max_tries):
        l, info = torch.linalg.cholesky_ex(k + jitter0 * (10.0 ** i) * eye)
        if int(info) == 0:
            return l
    raise RuntimeError("matrix not positive definite")


def gp_logp(e, y_c, sv, mean, noise, jitter0=None):
    """One ExactGP model: log N(y | mean, sv E + noise I), Cholesky route (N <= max_cholesky_size)."""
    n = e.shape[0]
    k = sv * e + noise * torch.eye(n, dtype=e.dtype)
    l = psd_safe_cholesky(k, jitter0)
    r = (y_c - mean).unsqueeze(-1)
    alpha = torch.cholesky_solve(r, l)
    quad = (r * alpha).sum()
    logdet = 2.0 * torch.log(torch.diagonal(l)).sum()
    return -0.5 * (quad + logdet + n * LOG_2PI), alpha.squeeze(-1)


def _per_model(p, c):
    """Base-kernel parameter of class model c: every ExactGPLayer owns its parameters (methods/DKT.py:63-66, 352-370); a scalar
    or one-element tensor is shared."""
    if p is None or not torch.is_tensor(p) or p.numel() == 1:
        return p
    return p.reshape(-1)[c]


def classification_loss(z, n_way, outputscale, mean, noise, kernel="bncossim", lengthscale=None,
                        normalize=False, variance=1.0):
    """methods/DKT.py:141-162.  z: [N,D] backbone (incl. bn_out) output; returns (loss, logp[C], alpha[C,N])."""
    if normalize:
        z = F.normalize(z, p=2, dim=1)
    n = z.shape[0]
    y = one_vs_rest_targets(n_way, n // n_way, z.dtype)
    logps, alphas = [], []
    for c in range(n_way):                       # IndependentModelList: a Python loop over models,
        e = base_matrix(z, None, kernel, _per_model(lengthscale, c))   # each evaluating its OWN kernel (own lengthscale / offset)
        lp, a = gp_logp(e, y[c], outputscale[c] * _per_model(variance, c), mean[c], noise[c])
        logps.append(lp / n)                      # ExactMarginalLogLikelihood: / num_data
        alphas.append(a)
    loss = -(sum(logps) / n_way)                  # SumMarginalLogLikelihood: / len(mlls)
    return loss, torch.stack(logps) * n, torch.stack(alphas)


def regression_loss(z, labels, outputscale, mean, noise, lengthscale, kernel="rbf"):
    """methods/DKT_regression.py:50-54."""
    n = z.shape[0]
    e = base_matrix(z, None, kernel, lengthscale)
    lp, a = gp_logp(e, labels, outputscale, mean, noise)
    return -lp / n, lp, a


def predict_mean(z_cond, z_star, alpha, outputscale, mean, kernel="bncossim", lengthscale=None, variance=1.0):
    rows = []
    for c in range(alpha.shape[0]):
        ex = base_matrix(z_star, z_cond, kernel, _per_model(lengthscale, c))
        rows.append(mean[c] + outputscale[c] * _per_model(variance, c) * (ex @ alpha[c]))
    return torch.stack(rows)


# --------------------------------------------------------------------------------------------
# fp32 CPU baseline ("port"): one training episode forward + backward, as the reference runs it
# --------------------------------------------------------------------------------------------
def cpu_baseline_train_episode(z_raw, n_way, raw_outputscale, mean, noise_value=0.1):
    """z_raw: fp32 [N,D] leaf (requires_grad) standing for the bn_out output; does F.normalize,
    the per-class GP loop, loss, and loss.backward().  Returns the loss tensor (detached)."""
    outputscale = F.softplus(raw_outputscale)
    noise = torch.full((n_way,), noise_value, dtype=z_raw.dtype)
    loss, _, _ = classification_loss(z_raw, n_way, outputscale, mean, noise, "bncossim", normalize=True)
    loss.backward()
    return loss.detach()


def cpu_baseline_test_episode(z_support, z_query, n_way, raw_outputscale, mean, noise_value=0.1):
    """forward-only test episode (methods/DKT.py:236-272)."""
    with torch.no_grad():
        zs = F.normalize(z_support, p=2, dim=1)
        zq = F.normalize(z_query, p=2, dim=1)
        outputscale = F.softplus(raw_outputscale)
        noise = torch.full((n_way,), noise_value, dtype=zs.dtype)
        _, _, alpha = classification_loss(zs, n_way, outputscale, mean, noise, "bncossim")
        mu = predict_mean(zs, zq, alpha, outputscale, mean)
        return torch.sigmoid(mu).argmax(0)
