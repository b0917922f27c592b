"""CPU oracle for the DKT hot path -- TEST INFRASTRUCTURE ONLY.

    *** PARITY UNPINNED ***
    The arithmetic of this path lives in GPyTorch (third-party, not vendored under
    /root/reference, prose-pinned to gpytorch 1.0.1 in reference README.md:27,37), which is
    not installed and not installable in this image, and the reference ships no tests,
    golden vectors or fixtures for it.  This file therefore RESTATES
    the published algorithm in float64 numpy and is validated mathematically against
    scikit-learn's GaussianProcessRegressor and scipy.stats.multivariate_normal
    (tests/test_oracle.py) -- not against outputs of the reference itself.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (deep-kernel-transfer_amd/) never does.

Reference call sites restated here (file:line under /root/reference):
  methods/DKT.py:129-136, 227-234   one-vs-rest +-1 targets            -> one_vs_rest_targets
  methods/DKT.py:45-48, 141-142     bn_out (BatchNorm1d) + F.normalize -> batchnorm1d_*, l2_normalize
  methods/DKT.py:337-378            ExactGPLayer (ConstantMean, ScaleKernel(LinearKernel|RBFKernel),
                                    noise fixed 0.1)                   -> gram_linear, gram_rbf, GPHypers
  methods/DKT.py:160-162            loss = -SumMLL(output, targets)    -> classification_loss
  methods/DKT.py:170-192, 258-272   eval-mode posterior mean, sigmoid, argmax -> posterior_mean, classify
  methods/DKT.py:297-335            get_logits                         -> posterior_mean (stacked [M, C])
  methods/DKT_regression.py:45-64   loss = -ExactMLL(pred, targets)    -> regression_loss
  methods/DKT_regression.py:66-97   condition on support, predict all, MSE -> regression_predict
  methods/DKT_regression.py:121-122 SpectralMixtureKernel(num_mixtures=4, ard_num_dims=2916) -> gram_spectral_mixture
GPyTorch 1.0.1 semantics restated (from its published source, recalled):
  utils/cholesky.py psd_safe_cholesky  : try plain Cholesky, then total diagonal jitter
                                         1e-6, 1e-5, 1e-4 (fp32) / 1e-8.. (fp64), 3 tries.
  distributions/multivariate_normal.py : log_prob = -0.5 (r^T K^-1 r + logdet K + N log 2 pi)
  mlls/exact_marginal_log_likelihood.py: divides log_prob by the number of data points N
  mlls/sum_marginal_log_likelihood.py  : mean over the n_way models
  constraints: Positive() = softplus(raw); GaussianLikelihood noise = softplus(raw) + 1e-4
  kernels/rbf_kernel.py + kernel.py    : x/lengthscale, squared distance through the centred
                                         norm expansion, clamp >= 0, exp(-d2/2)
  kernels/spectral_mixture_kernel.py   : k = sum_q w_q prod_d exp(-2 pi^2 (x1 s_qd - x2 s_qd)^2) cos(2 pi (x1 m_qd - x2 m_qd));
                                         weights [Q], means / scales [Q,1,D], all softplus(raw), raw init 0; no ScaleKernel
  models/exact_prediction_strategies.py: mean_cache = (K + s2 I)^-1 (y - m);  mu* = m + K*^T mean_cache;
                                         cov* = K** - K*^T (K + s2 I)^-1 K*  (+ s2 I through the likelihood)
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------------------------------------
# constraints (gpytorch/constraints/constraints.py: Positive -> softplus, GreaterThan(1e-4))
# ----------------------------------------------------------------------------------------------
def softplus(x):
    x = np.asarray(x, dtype=np.float64)
    return np.where(x > 30.0, x, np.log1p(np.exp(np.minimum(x, 30.0))))


def inv_softplus(y):
    y = np.asarray(y, dtype=np.float64)
    return np.where(y > 30.0, y, np.log(np.expm1(np.minimum(y, 30.0))))


def sigmoid(x):
    x = np.asarray(x, dtype=np.float64)
    return 1.0 / (1.0 + np.exp(-x))


NOISE_LOWER_BOUND = 1e-4  # GaussianLikelihood default noise_constraint = GreaterThan(1e-4)


# ----------------------------------------------------------------------------------------------
# front half: bn_out + L2 normalise  (methods/DKT.py:45-48, 141-142, 174-175)
# ----------------------------------------------------------------------------------------------
def batchnorm1d_train(z, gamma=None, beta=None, eps=1e-5):
    """nn.BatchNorm1d in train mode: per-episode batch mean, BIASED variance.
    Returns (out, batch_mean, unbiased_var) -- the last two feed the running-stat update
    (momentum 0.1, unbiased variance, as torch does)."""
    z = np.asarray(z, dtype=np.float64)
    mu = z.mean(axis=0)
    var_b = z.var(axis=0)  # biased
    out = (z - mu) / np.sqrt(var_b + eps)
    if gamma is not None:
        out = out * gamma
    if beta is not None:
        out = out + beta
    n = z.shape[0]
    var_u = var_b * n / max(n - 1, 1)
    return out, mu, var_u


def batchnorm1d_eval(z, running_mean, running_var, gamma=None, beta=None, eps=1e-5):
    z = np.asarray(z, dtype=np.float64)
    out = (z - running_mean) / np.sqrt(running_var + eps)
    if gamma is not None:
        out = out * gamma
    if beta is not None:
        out = out + beta
    return out


def l2_normalize(z, eps=1e-12):
    """F.normalize(z, p=2, dim=1): z_i / max(||z_i||_2, eps)."""
    z = np.asarray(z, dtype=np.float64)
    nrm = np.sqrt((z * z).sum(axis=1, keepdims=True))
    return z / np.maximum(nrm, eps)


def one_vs_rest_targets(n_way, per_class):
    """methods/DKT.py:129-136: Y[c, i] = +1 on rows [c*spm, (c+1)*spm), else -1 (class-major rows)."""
    n = n_way * per_class
    y = -np.ones((n_way, n), dtype=np.float64)
    for c in range(n_way):
        y[c, c * per_class:(c + 1) * per_class] = 1.0
    return y


# ----------------------------------------------------------------------------------------------
# kernels (base kernel matrix E; the ScaleKernel outputscale is applied in the MLL)
# ----------------------------------------------------------------------------------------------
def gram_linear(za, zb=None):
    """LinearKernel with variance 1 (cossim / bncossim freeze variance = 1, DKT.py:366-370)."""
    za = np.asarray(za, dtype=np.float64)
    zb = za if zb is None else np.asarray(zb, dtype=np.float64)
    return za @ zb.T


def sq_dist(za, zb=None):
    za = np.asarray(za, dtype=np.float64)
    zb = za if zb is None else np.asarray(zb, dtype=np.float64)
    d2 = ((za[:, None, :] - zb[None, :, :]) ** 2).sum(-1) if za.shape[0] * zb.shape[0] * za.shape[1] < 5e7 \
        else (za * za).sum(1)[:, None] + (zb * zb).sum(1)[None, :] - 2.0 * za @ zb.T
    return np.maximum(d2, 0.0)


def gram_rbf(za, zb=None, lengthscale=1.0):
    """RBFKernel: exp(-0.5 * ||(a - b)/l||^2)."""
    return np.exp(-0.5 * sq_dist(za, zb) / (lengthscale * lengthscale))


def gram_matern25(za, zb=None, lengthscale=1.0):
    """MaternKernel(nu=2.5) (gpytorch default nu): (1 + sqrt5 d + 5/3 d^2) exp(-sqrt5 d), d = ||a-b||/l.
    (GPyTorch clamps d^2 >= 1e-30 before the sqrt.)"""
    d = np.sqrt(np.maximum(sq_dist(za, zb), 1e-30)) / lengthscale
    s5 = math.sqrt(5.0)
    return (1.0 + s5 * d + (5.0 / 3.0) * d * d) * np.exp(-s5 * d)


def gram_poly(za, zb=None, power=1, offset=0.0):
    """PolynomialKernel(power): (a.b + offset)^power, offset = softplus(raw_offset)."""
    return (gram_linear(za, zb) + offset) ** power


def gram_spectral_mixture(za, zb, weights, means, scales, terms=False):
    """SpectralMixtureKernel: sum_q w_q prod_d exp(-2 pi^2 (sigma_qd tau_d)^2) cos(2 pi mu_qd tau_d), tau = a - b.
    weights [Q], means / scales [Q,D] (constrained values).  terms=True also returns the per-mixture matrices [Q,M,N]."""
    za = np.asarray(za, dtype=np.float64)
    zb = za if zb is None else np.asarray(zb, dtype=np.float64)
    w = np.asarray(weights, dtype=np.float64).reshape(-1)
    mu = np.asarray(means, dtype=np.float64).reshape(w.size, -1)
    sg = np.asarray(scales, dtype=np.float64).reshape(w.size, -1)
    tau = za[:, None, :] - zb[None, :, :]                                   # [M,N,D]
    eq = np.empty((w.size, za.shape[0], zb.shape[0]))
    for q in range(w.size):
        eq[q] = (np.exp(-2.0 * math.pi ** 2 * (tau * sg[q]) ** 2) * np.cos(2.0 * math.pi * tau * mu[q])).prod(-1)
    e = np.tensordot(w, eq, axes=(0, 0))
    return (e, eq) if terms else e


# ----------------------------------------------------------------------------------------------
# jittered Cholesky (gpytorch/utils/cholesky.py @ 1.0.1)
# ----------------------------------------------------------------------------------------------
class NotPSDError(RuntimeError):
    pass


def psd_safe_cholesky(k, jitter0=1e-6, max_tries=3):
    """Returns (L, jitter_used).  Try 0 jitter first; on failure TOTAL diagonal jitter
    jitter0 * 10**i for i = 0..max_tries-1.  Failure = LAPACK potrf failure
    (a non-positive or NaN pivot)."""
    k = np.asarray(k, dtype=np.float64)
    try:
        return np.linalg.cholesky(k), 0.0
    except np.linalg.LinAlgError:
        pass
    if np.isnan(k).any():
        raise NotPSDError("NaN in matrix")
    for i in range(max_tries):
        jit = jitter0 * (10.0 ** i)
        try:
            return np.linalg.cholesky(k + jit * np.eye(k.shape[0])), jit
        except np.linalg.LinAlgError:
            continue
    raise NotPSDError("matrix not positive definite after %d jitter tries" % max_tries)


def _solve_lower(l, b):
    import scipy.linalg as sla
    return sla.solve_triangular(l, b, lower=True)


def _solve_upper_t(l, b):
    import scipy.linalg as sla
    return sla.solve_triangular(l.T, b, lower=False)


# ----------------------------------------------------------------------------------------------
# exact-GP marginal log likelihood for C one-vs-rest models sharing one base matrix E
# ----------------------------------------------------------------------------------------------
@dataclass
class MLLResult:
    logp: np.ndarray        # [C]   un-normalised log N(y_c | m_c, K_c)
    alpha: np.ndarray       # [C,N] mean cache (K_c + s2 I)^-1 (y_c - m_c)
    chol: np.ndarray        # [C,N,N] lower Cholesky factors
    jitter: np.ndarray      # [C]
    quad: np.ndarray        # [C]   r^T K^-1 r
    logdet: np.ndarray      # [C]   log det K_c


def mll_terms(e, y, sv, mean, noise, jitter0=1e-6, max_tries=3):
    """e: [N,N] base kernel matrix; y: [C,N]; sv,mean,noise: [C] (outputscale*variance, constant mean,
    likelihood noise).  K_c = sv_c E + noise_c I (ScaleKernel + GaussianLikelihood.marginal)."""
    e = np.asarray(e, dtype=np.float64)
    y = np.atleast_2d(np.asarray(y, dtype=np.float64))
    c_, n = y.shape
    sv = np.broadcast_to(np.asarray(sv, dtype=np.float64), (c_,))
    mean = np.broadcast_to(np.asarray(mean, dtype=np.float64), (c_,))
    noise = np.broadcast_to(np.asarray(noise, dtype=np.float64), (c_,))
    logp = np.zeros(c_)
    alpha = np.zeros((c_, n))
    chol = np.zeros((c_, n, n))
    jit = np.zeros(c_)
    quad = np.zeros(c_)
    logdet = np.zeros(c_)
    for c in range(c_):  # the reference's per-class Python loop (IndependentModelList)
        k = sv[c] * e + noise[c] * np.eye(n)
        l, j = psd_safe_cholesky(k, jitter0, max_tries)
        r = y[c] - mean[c]
        w = _solve_lower(l, r)
        a = _solve_upper_t(l, w)
        quad[c] = float(w @ w)
        logdet[c] = 2.0 * float(np.log(np.diag(l)).sum())
        logp[c] = -0.5 * (quad[c] + logdet[c] + n * LOG_2PI)
        alpha[c] = a
        chol[c] = l
        jit[c] = j
    return MLLResult(logp, alpha, chol, jit, quad, logdet)


def classification_loss(logp, n):
    """methods/DKT.py:162 with SumMarginalLogLikelihood(ExactMarginalLogLikelihood):
    loss = -(1/C) sum_c logp_c / N."""
    logp = np.asarray(logp, dtype=np.float64)
    return -float(np.mean(logp / n))


def regression_loss(logp, n):
    """methods/DKT_regression.py:54: loss = -logp / N."""
    return -float(np.asarray(logp).reshape(-1)[0]) / n


def mll_grads(e, res: MLLResult, sv, noise, weight):
    """Closed-form gradient of  obj = sum_c weight_c * logp_c  w.r.t. E, sv_c, mean_c, noise_c.
    dlogp/dK = 0.5 (alpha alpha^T - K^-1)."""
    e = np.asarray(e, dtype=np.float64)
    c_, n = res.alpha.shape
    sv = np.broadcast_to(np.asarray(sv, dtype=np.float64), (c_,))
    weight = np.broadcast_to(np.asarray(weight, dtype=np.float64), (c_,))
    w_e = np.zeros((n, n))
    dsv = np.zeros(c_)
    dmean = np.zeros(c_)
    dnoise = np.zeros(c_)
    eye = np.eye(n)
    for c in range(c_):
        linv = _solve_lower(res.chol[c], eye)
        kinv = linv.T @ linv
        m = 0.5 * (np.outer(res.alpha[c], res.alpha[c]) - kinv)
        w_e += weight[c] * sv[c] * m
        dsv[c] = weight[c] * float((m * e).sum())
        dmean[c] = weight[c] * float(res.alpha[c].sum())
        dnoise[c] = weight[c] * float(np.trace(m))
    return w_e, dsv, dmean, dnoise


def gram_linear_bwd(w_e, z):
    """d obj / d Z for E = Z Z^T: (W + W^T) Z."""
    return (w_e + w_e.T) @ np.asarray(z, dtype=np.float64)


def gram_rbf_bwd(w_e, e, z, lengthscale):
    """d obj / d Z and d obj / d lengthscale for E = exp(-0.5 d2 / l^2)."""
    z = np.asarray(z, dtype=np.float64)
    ws = 0.5 * (w_e + w_e.T)
    a = -(ws * e) / (lengthscale ** 2)
    wp = np.diag(a.sum(1)) - a
    dz = 2.0 * wp @ z
    d2 = sq_dist(z)
    dl = float((ws * e * d2).sum()) / lengthscale ** 3
    return dz, dl


# ----------------------------------------------------------------------------------------------
# prediction (eval-mode ExactGP + likelihood)
# ----------------------------------------------------------------------------------------------
def posterior_mean(e_star, alpha, sv, mean):
    """e_star: [M,N] base cross kernel k(x*, X); returns mu [C,M] = m_c + sv_c E* alpha_c."""
    e_star = np.asarray(e_star, dtype=np.float64)
    alpha = np.atleast_2d(alpha)
    c_ = alpha.shape[0]
    sv = np.broadcast_to(np.asarray(sv, dtype=np.float64), (c_,))
    mean = np.broadcast_to(np.asarray(mean, dtype=np.float64), (c_,))
    return mean[:, None] + sv[:, None] * (alpha @ e_star.T)


def posterior_var(e_star, e_star_diag, chol, sv, noise, add_noise=True):
    """Diagonal of the predictive covariance per class: sv k** - sv^2 k*^T K^-1 k* (+ noise)."""
    e_star = np.asarray(e_star, dtype=np.float64)
    c_ = chol.shape[0]
    sv = np.broadcast_to(np.asarray(sv, dtype=np.float64), (c_,))
    noise = np.broadcast_to(np.asarray(noise, dtype=np.float64), (c_,))
    out = np.zeros((c_, e_star.shape[0]))
    for c in range(c_):
        v = _solve_lower(chol[c], sv[c] * e_star.T)  # [N,M]
        out[c] = sv[c] * np.asarray(e_star_diag) - (v * v).sum(0)
        if add_noise:
            out[c] += noise[c]
    return out


def classify(mu):
    """methods/DKT.py:267-269: argmax over classes of sigmoid(mean) (first max wins, np.argmax)."""
    return np.argmax(sigmoid(mu), axis=0)


# ----------------------------------------------------------------------------------------------
# whole-episode restatements
# ----------------------------------------------------------------------------------------------
@dataclass
class GPHypers:
    """Constrained values.  Defaults = the reference's initial state: raw_outputscale 0 -> ln 2,
    constant mean 0, classification noise forced to 0.1 and frozen (DKT.py:346-347), regression noise
    softplus(0) + 1e-4 and learned, lengthscale softplus(0) = ln 2."""
    outputscale: np.ndarray
    mean: np.ndarray
    noise: np.ndarray
    lengthscale: float = math.log(2.0)
    variance: float = 1.0
    mixture: tuple = None      # spectral kernel only: (weights [Q], means [Q,D], scales [Q,D]), constrained values

    @staticmethod
    def init_classification(n_way):
        return GPHypers(np.full(n_way, math.log(2.0)), np.zeros(n_way), np.full(n_way, 0.1))

    @staticmethod
    def init_regression():
        return GPHypers(np.full(1, math.log(2.0)), np.zeros(1), np.full(1, math.log(2.0) + NOISE_LOWER_BOUND))


def base_matrix(za, zb, kernel, hyp: GPHypers):
    if kernel in ("linear", "cossim", "bncossim"):
        return gram_linear(za, zb)
    if kernel in ("rbf", "RBF"):
        return gram_rbf(za, zb, hyp.lengthscale)
    if kernel == "matern":
        return gram_matern25(za, zb, hyp.lengthscale)
    if kernel == "poli1":
        return gram_poly(za, zb, 1, hyp.lengthscale)  # offset carried in .lengthscale slot
    if kernel == "poli2":
        return gram_poly(za, zb, 2, hyp.lengthscale)
    if kernel == "spectral":
        return gram_spectral_mixture(za, zb, *hyp.mixture)
    raise ValueError("[ERROR] the kernel '" + str(kernel) + "' is not supported!")


def train_episode(z, n_way, hyp: GPHypers, kernel="bncossim", jitter0=1e-6):
    """methods/DKT.py:129-162 after the backbone: z [N,D] is ALREADY bn_out'ed + normalised.
    Returns dict(loss, logp, alpha, chol, grads...)."""
    z = np.asarray(z, dtype=np.float64)
    n = z.shape[0]
    y = one_vs_rest_targets(n_way, n // n_way)
    e = base_matrix(z, None, kernel, hyp)
    sv = hyp.outputscale * hyp.variance
    res = mll_terms(e, y, sv, hyp.mean, hyp.noise, jitter0)
    loss = classification_loss(res.logp, n)
    weight = np.full(n_way, -1.0 / (n_way * n))
    w_e, dsv, dmean, dnoise = mll_grads(e, res, sv, hyp.noise, weight)
    out = dict(loss=loss, logp=res.logp, alpha=res.alpha, chol=res.chol, jitter=res.jitter, e=e, y=y,
               w_e=w_e, dsv=dsv, dmean=dmean, dnoise=dnoise)
    if kernel in ("linear", "cossim", "bncossim"):
        out["dz"] = gram_linear_bwd(w_e, z)
    elif kernel in ("rbf", "RBF"):
        out["dz"], out["dlengthscale"] = gram_rbf_bwd(w_e, e, z, hyp.lengthscale)
    return out


def eval_episode(z_support, z_query, n_way, hyp: GPHypers, kernel="bncossim", jitter0=1e-6):
    """methods/DKT.py:224-272 (correct) / 297-335 (get_logits) after the backbone."""
    zs = np.asarray(z_support, dtype=np.float64)
    zq = np.asarray(z_query, dtype=np.float64)
    ns = zs.shape[0]
    y = one_vs_rest_targets(n_way, ns // n_way)
    e = base_matrix(zs, None, kernel, hyp)
    sv = hyp.outputscale * hyp.variance
    res = mll_terms(e, y, sv, hyp.mean, hyp.noise, jitter0)
    ex = base_matrix(zq, zs, kernel, hyp)
    mu = posterior_mean(ex, res.alpha, sv, hyp.mean)
    labels = classify(mu)
    y_query = np.repeat(np.arange(n_way), zq.shape[0] // n_way)
    return dict(mu=mu, labels=labels, correct=float((labels == y_query).sum()), count=len(y_query),
                logits=mu.T.copy(), alpha=res.alpha, logp=res.logp)


def regression_episode(z, labels, hyp: GPHypers, kernel="rbf", jitter0=1e-6):
    """methods/DKT_regression.py:50-56 after the backbone (single GP, noise learned)."""
    z = np.asarray(z, dtype=np.float64)
    n = z.shape[0]
    e = base_matrix(z, None, kernel, hyp)
    sv = hyp.outputscale * hyp.variance
    res = mll_terms(e, np.asarray(labels, dtype=np.float64)[None, :], sv, hyp.mean, hyp.noise, jitter0)
    loss = regression_loss(res.logp, n)
    weight = np.full(1, -1.0 / n)
    w_e, dsv, dmean, dnoise = mll_grads(e, res, sv, hyp.noise, weight)
    out = dict(loss=loss, logp=res.logp, alpha=res.alpha, chol=res.chol, e=e,
               w_e=w_e, dsv=dsv, dmean=dmean, dnoise=dnoise, prior_mean=np.full(n, hyp.mean[0]))
    if kernel in ("rbf", "RBF"):
        out["dz"], out["dlengthscale"] = gram_rbf_bwd(w_e, e, z, hyp.lengthscale)
    return out


def regression_predict(z_support, y_support, z_all, hyp: GPHypers, kernel="rbf", jitter0=1e-6):
    """methods/DKT_regression.py:83-95: condition on support, predict all frames, MSE vs all."""
    zs = np.asarray(z_support, dtype=np.float64)
    za = np.asarray(z_all, dtype=np.float64)
    e = base_matrix(zs, None, kernel, hyp)
    sv = hyp.outputscale * hyp.variance
    res = mll_terms(e, np.asarray(y_support, dtype=np.float64)[None, :], sv, hyp.mean, hyp.noise, jitter0)
    ex = base_matrix(za, zs, kernel, hyp)
    mu = posterior_mean(ex, res.alpha, sv, hyp.mean)[0]
    if kernel == "spectral":
        exx_diag = np.full(za.shape[0], float(np.sum(hyp.mixture[0])))      # tau = 0: every mixture term is 1
    else:
        exx_diag = np.ones(za.shape[0]) if kernel in ("rbf", "RBF", "matern") else (za * za).sum(1)
    var = posterior_var(ex, exx_diag, res.chol, sv, hyp.noise, add_noise=True)[0]
    return dict(mean=mu, var=var, lower=mu - 2.0 * np.sqrt(var), upper=mu + 2.0 * np.sqrt(var))


# ----------------------------------------------------------------------------------------------
# synthetic workloads
# ----------------------------------------------------------------------------------------------
def synthetic_features(b, n, d, seed, correlated_n_way=0):
    """Zraw ~ N(0,1) -> BN1d(train, gamma=1, beta=0) -> L2 normalise, per episode.  float64 [B,N,D].
    correlated_n_way > 0: Z = 0.9 class_mean + 0.1 noise before BN (cond(K) ~ 1e2..1e3)."""
    rng = np.random.default_rng(seed)
    out = np.empty((b, n, d))
    for i in range(b):
        zr = rng.standard_normal((n, d))
        if correlated_n_way:
            cm = rng.standard_normal((correlated_n_way, d))
            zr = 0.9 * np.repeat(cm, n // correlated_n_way, axis=0) + 0.1 * zr
        zb, _, _ = batchnorm1d_train(zr)
        out[i] = l2_normalize(zb)
    return out


def perturbed_hypers(n_way, seed):
    rng = np.random.default_rng(seed)
    return GPHypers(softplus(rng.normal(0.0, 0.5, n_way)), rng.normal(0.0, 0.1, n_way), np.full(n_way, 0.1))
