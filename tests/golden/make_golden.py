"""Generates tests/golden/*.npz -- small seeded input/expected-output vectors for the DKT hot path.

PARITY UNPINNED: the reference's own implementation of this path cannot run here (GPyTorch is not
installed / installable), so the expected values come from the float64 oracle
(oracle/dkt_oracle.py), cross-checked at generation time against scikit-learn's
GaussianProcessRegressor and scipy.stats.multivariate_normal.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dkt_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# (name, n_way, n_support, n_query, D, kernel, correlated)
TRAIN_CASES = [
    ("cfg1_omniglot_5w5s_conv4s", 5, 5, 16, 64, "bncossim", 0),
    ("cfg2_cub_5w5s_conv4", 5, 5, 16, 1600, "bncossim", 0),
    ("cfg2_cub_5w5s_conv4_corr", 5, 5, 16, 1600, "bncossim", 5),
    ("cfg3_mini_5w1s_resnet10", 5, 1, 16, 512, "bncossim", 0),
    ("cfg4_mini_20w5s_resnet18", 20, 5, 16, 512, "bncossim", 0),
    ("small_3w2s_rbf", 3, 2, 3, 20, "rbf", 0),
]
TEST_CASES = [
    ("test_5w5s_d64", 5, 5, 15, 64),
    ("test_5w1s_d512", 5, 1, 15, 512),
    ("test_20w5s_d512", 20, 5, 15, 512),
]


def sklearn_logp(e_kernel, z, y, sv, noise, ls=None):
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import ConstantKernel as C, DotProduct, WhiteKernel, RBF
    base = DotProduct(sigma_0=0.0, sigma_0_bounds="fixed") if e_kernel == "linear" else RBF(length_scale=ls)
    k = C(sv, "fixed") * base + WhiteKernel(noise, "fixed")
    gp = GaussianProcessRegressor(kernel=k, alpha=0.0, optimizer=None).fit(z, y)
    return gp.log_marginal_likelihood_value_, gp


def main():
    for name, c, s, q, d, kernel, corr in TRAIN_CASES:
        n = c * (s + q)
        seed = abs(hash(name)) % 1000 + 11
        seed = sum(ord(ch) for ch in name)           # stable across processes
        z = O.synthetic_features(1, n, d, seed, corr)[0]
        if kernel == "rbf":
            z = z * 3.0 + 0.5
        hyp = O.perturbed_hypers(c, seed + 1)
        if kernel == "rbf":
            hyp.lengthscale = 1.3
        out = O.train_episode(z, c, hyp, kernel)
        # cross-check class 0 against sklearn + scipy
        from scipy.stats import multivariate_normal
        kmat = hyp.outputscale[0] * out["e"] + hyp.noise[0] * np.eye(n)
        ref_scipy = multivariate_normal.logpdf(out["y"][0], mean=np.full(n, hyp.mean[0]), cov=kmat)
        ref_sk, _ = sklearn_logp("linear" if kernel != "rbf" else "rbf", z, out["y"][0] - hyp.mean[0],
                                 hyp.outputscale[0], hyp.noise[0], hyp.lengthscale)
        assert abs(ref_scipy - out["logp"][0]) < 1e-8 * abs(ref_scipy), (name, ref_scipy, out["logp"][0])
        assert abs(ref_sk - out["logp"][0]) < 1e-8 * abs(ref_sk), (name, ref_sk, out["logp"][0])
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            seed=seed, n_way=c, n_support=s, n_query=q, d=d, kernel=kernel, correlated=corr,
            z_scale=(3.0 if kernel == "rbf" else 1.0), z_shift=(0.5 if kernel == "rbf" else 0.0),
            z=(z.astype(np.float64) if z.size <= 12000 else np.zeros(0)),   # big inputs are regenerated from the seed
            z_checksum=float(z.sum()), z_row0=z[0, :8].copy(),
            outputscale=hyp.outputscale, mean=hyp.mean, noise=hyp.noise, lengthscale=hyp.lengthscale,
            loss=out["loss"], logp=out["logp"], logp0_sklearn=ref_sk, logp0_scipy=ref_scipy,
            alpha=out["alpha"], chol_diag=np.stack([np.diag(l) for l in out["chol"]]),
            dsv=out["dsv"], dmean=out["dmean"], dnoise=out["dnoise"],
            dz_checksum=float(np.abs(out["dz"]).sum()), dz_rows=out["dz"][:3].copy(), dz_fro=float(np.linalg.norm(out["dz"])),
            dlengthscale=out.get("dlengthscale", 0.0), w_e_fro=float(np.linalg.norm(out["w_e"])),
        )
        print("wrote", name, "loss", out["loss"])
    for name, c, s, q, d in TEST_CASES:
        seed = sum(ord(ch) for ch in name)
        zall = O.synthetic_features(1, c * (s + q), d, seed, c)[0].reshape(c, s + q, d)
        zs = zall[:, :s].reshape(c * s, d)
        zq = zall[:, s:].reshape(c * q, d)
        hyp = O.perturbed_hypers(c, seed + 1)
        out = O.eval_episode(zs, zq, c, hyp)
        _, gp = sklearn_logp("linear", zs, O.one_vs_rest_targets(c, s)[0] - hyp.mean[0], hyp.outputscale[0], hyp.noise[0])
        mu0 = gp.predict(zq) + hyp.mean[0]
        assert np.abs(mu0 - out["mu"][0]).max() < 1e-8
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"), seed=seed, n_way=c, n_support=s, n_query=q, d=d,
            z_checksum=float(zall.sum()), outputscale=hyp.outputscale, mean=hyp.mean, noise=hyp.noise,
            mu=out["mu"], labels=out["labels"], correct=out["correct"], count=out["count"], alpha=out["alpha"], logp=out["logp"],
        )
        print("wrote", name, "acc", out["correct"] / out["count"])
    # degenerate: duplicated rows (rank deficient, relies on noise) and near-singular needing jitter
    rng = np.random.default_rng(5)
    z = O.l2_normalize(rng.standard_normal((12, 16)))
    z[6:] = z[:6]                                   # exact duplicates
    e = O.gram_linear(z)
    y = O.one_vs_rest_targets(2, 6)
    res = O.mll_terms(e, y, np.array([0.7, 1.1]), np.array([0.05, -0.02]), np.array([0.1, 0.1]))
    tiny = O.mll_terms(e, y, np.array([1.0, 1.0]), np.zeros(2), np.zeros(2), jitter0=1e-6)   # singular -> jitter
    np.savez_compressed(os.path.join(OUT, "degenerate.npz"), z=z, logp_dup=res.logp, alpha_dup=res.alpha,
                        jitter_dup=res.jitter, logp_sing=tiny.logp, jitter_sing=tiny.jitter)
    print("wrote degenerate; jitter used for the singular case:", tiny.jitter)


def make_spectral():
    """Regression head with the SpectralMixture kernel (DKT_regression.py:121-122): numpy restatement cross-checked against
    an independent torch float64 formulation (whose autograd supplies the expected gradients) and scipy's logpdf."""
    import torch
    from scipy.stats import multivariate_normal
    from oracle import dkt_oracle_torch as T
    rng = np.random.default_rng(11)
    n, d, q = 19, 48, 4
    z = rng.standard_normal((n, d)) * 0.15
    labels = rng.uniform(-1.0, 1.0, n)
    w, mu, sg = rng.random(q) * 0.6 + 0.3, rng.random((q, d)) * 0.6 + 0.1, rng.random((q, d)) * 0.6 + 0.1
    hyp = O.GPHypers(np.ones(1), np.array([0.1]), np.array([0.25]), mixture=(w, mu, sg))
    out = O.regression_episode(z, labels, hyp, kernel="spectral")
    ref = multivariate_normal(mean=np.full(n, 0.1), cov=out["e"] + 0.25 * np.eye(n)).logpdf(labels)
    assert abs(ref - out["logp"][0]) < 1e-9 * abs(ref)
    t = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (z, w, mu, sg)]
    e_t = T.spectral_mixture(t[0], None, t[1], t[2], t[3])
    assert np.abs(e_t.detach().numpy() - out["e"]).max() < 1e-13
    lp, _ = T.gp_logp(e_t, torch.tensor(labels), torch.ones((), dtype=torch.float64), torch.tensor(0.1, dtype=torch.float64),
                      torch.tensor(0.25, dtype=torch.float64))
    (-lp / n).backward()
    sup = [0, 3, 7, 11, 18]
    pred = O.regression_predict(z[sup], labels[sup], z, hyp, kernel="spectral")
    np.savez_compressed(os.path.join(OUT, "regression_spectral_q4.npz"), z=z, labels=labels, weights=w, means=mu, scales=sg,
                        mean=hyp.mean, noise=hyp.noise, e=out["e"], logp=out["logp"], loss=out["loss"], alpha=out["alpha"],
                        dz=t[0].grad.numpy(), dweights=t[1].grad.numpy(), dmeans=t[2].grad.numpy(), dscales=t[3].grad.numpy(),
                        dmean=out["dmean"], dnoise=out["dnoise"], support=np.array(sup), pred_mean=pred["mean"], pred_var=pred["var"],
                        logp_scipy=ref)
    print("wrote regression_spectral_q4: loss", out["loss"])


def make_cfg0():
    """BASELINE.json configs[0] / SURVEY.md 8c: the QMUL regression head (methods/DKT_regression.py:45-97) at its real shape --
    one GP, RBF kernel, a training task of 19 frames with D = 2916 Conv3 features, and the test-time shape (condition on 5 of
    them, predict all 19).  Features are regenerated from the seed (19 x 2916 doubles would be 440 KB).  Cross-checked against
    scikit-learn (log marginal likelihood, predictive mean / std) and scipy at generation time."""
    from scipy.stats import multivariate_normal
    seed, n, d = 1907, 19, 2916
    rng = np.random.default_rng(seed)
    z = np.abs(rng.standard_normal((n, d))) * 0.35 + 0.05 * rng.standard_normal((1, d))     # ReLU-like Conv3 outputs
    labels = rng.uniform(-1.0, 1.0, n)
    hyp = O.GPHypers(np.array([0.9]), np.array([0.07]), np.array([0.3]), lengthscale=11.0)
    out = O.regression_episode(z, labels, hyp, kernel="rbf")
    kmat = hyp.outputscale[0] * out["e"] + hyp.noise[0] * np.eye(n)
    ref_scipy = multivariate_normal.logpdf(labels, mean=np.full(n, hyp.mean[0]), cov=kmat)
    ref_sk, _ = sklearn_logp("rbf", z, labels - hyp.mean[0], hyp.outputscale[0], hyp.noise[0], hyp.lengthscale)
    assert abs(ref_scipy - out["logp"][0]) < 1e-8 * abs(ref_scipy) and abs(ref_sk - out["logp"][0]) < 1e-8 * abs(ref_sk)
    sup = [1, 4, 9, 13, 17]
    pred = O.regression_predict(z[sup], labels[sup], z, hyp, kernel="rbf")
    _, gp = sklearn_logp("rbf", z[sup], labels[sup] - hyp.mean[0], hyp.outputscale[0], hyp.noise[0], hyp.lengthscale)
    mu_sk, sd_sk = gp.predict(z, return_std=True)
    assert np.abs(mu_sk + hyp.mean[0] - pred["mean"]).max() < 1e-8
    # sklearn's predictive variance takes the prior diagonal from the full kernel, WhiteKernel included: it is the variance
    # with the observation noise added -- what likelihood(model(x)).confidence_region() reads in the reference
    assert np.abs(sd_sk ** 2 - pred["var"]).max() < 1e-9
    np.savez_compressed(os.path.join(OUT, "cfg0_qmul_regression_rbf.npz"), seed=seed, n=n, d=d, z_checksum=float(z.sum()), z_row0=z[0, :8].copy(),
                        labels=labels, outputscale=hyp.outputscale, mean=hyp.mean, noise=hyp.noise, lengthscale=hyp.lengthscale,
                        e_rows=out["e"][:3].copy(), loss=out["loss"], logp=out["logp"], logp_sklearn=ref_sk, logp_scipy=ref_scipy,
                        alpha=out["alpha"], dsv=out["dsv"], dmean=out["dmean"], dnoise=out["dnoise"], dlengthscale=out["dlengthscale"],
                        dz_rows=out["dz"][:3].copy(), dz_fro=float(np.linalg.norm(out["dz"])), dz_checksum=float(np.abs(out["dz"]).sum()),
                        support=np.array(sup), pred_mean=pred["mean"], pred_var=pred["var"])
    print("wrote cfg0_qmul_regression_rbf: loss", out["loss"], "dlengthscale", out["dlengthscale"])


def cfg0_features(seed=1907, n=19, d=2916):
    """The generator the fixture's z comes from (tests regenerate z from the seed)."""
    rng = np.random.default_rng(seed)
    z = np.abs(rng.standard_normal((n, d))) * 0.35 + 0.05 * rng.standard_normal((1, d))
    return z, rng.uniform(-1.0, 1.0, n)


if __name__ == "__main__":
    if sys.argv[1:] == ["spectral"]:
        make_spectral()
    elif sys.argv[1:] == ["cfg0"]:
        make_cfg0()
    else:
        main()
        make_spectral()
