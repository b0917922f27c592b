"""Oracle self-checks (CPU): the float64 restatement against independent implementations
(scikit-learn GPR, scipy multivariate normal, torch autograd) and against the committed golden vectors.
PARITY UNPINNED -- see oracle/dkt_oracle.py header."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import dkt_oracle as O
from oracle import dkt_oracle_torch as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _regen(g):
    n = int(g["n_way"]) * (int(g["n_support"]) + int(g["n_query"]))
    z = O.synthetic_features(1, n, int(g["d"]), int(g["seed"]), int(g["correlated"]))[0]
    z = z * float(g["z_scale"]) + float(g["z_shift"])
    assert abs(z.sum() - float(g["z_checksum"])) < 1e-9
    np.testing.assert_allclose(z[0, :8], g["z_row0"], rtol=0, atol=1e-12)
    return z


def _hyp(g):
    return O.GPHypers(g["outputscale"], g["mean"], g["noise"], float(g["lengthscale"]))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "cfg[1-9]*.npz")) + glob.glob(os.path.join(GOLD, "small*.npz"))))
def test_golden_train_vectors(path):
    g = np.load(path)
    z = _regen(g)
    out = O.train_episode(z, int(g["n_way"]), _hyp(g), str(g["kernel"]))
    assert abs(out["loss"] - float(g["loss"])) < 1e-10
    np.testing.assert_allclose(out["logp"], g["logp"], rtol=1e-11)
    np.testing.assert_allclose(out["alpha"], g["alpha"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(out["dz"][:3], g["dz_rows"], rtol=1e-8, atol=1e-12)
    # independent implementations pinned at generation time
    assert abs(out["logp"][0] - float(g["logp0_sklearn"])) < 1e-8 * abs(out["logp"][0])
    assert abs(out["logp"][0] - float(g["logp0_scipy"])) < 1e-8 * abs(out["logp"][0])


def test_linear_mll_vs_sklearn_and_scipy():
    from scipy.stats import multivariate_normal
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import ConstantKernel as C, DotProduct, WhiteKernel
    z = O.synthetic_features(1, 105, 64, 3)[0]
    hyp = O.perturbed_hypers(5, 4)
    out = O.train_episode(z, 5, hyp)
    for c in range(5):
        k = C(hyp.outputscale[c], "fixed") * DotProduct(0.0, "fixed") + WhiteKernel(hyp.noise[c], "fixed")
        gp = GaussianProcessRegressor(kernel=k, alpha=0.0, optimizer=None).fit(z, out["y"][c] - hyp.mean[c])
        assert abs(gp.log_marginal_likelihood_value_ - out["logp"][c]) < 1e-9 * abs(out["logp"][c])
        kmat = hyp.outputscale[c] * out["e"] + hyp.noise[c] * np.eye(105)
        ref = multivariate_normal.logpdf(out["y"][c], mean=np.full(105, hyp.mean[c]), cov=kmat)
        assert abs(ref - out["logp"][c]) < 1e-9 * abs(ref)
    assert abs(out["loss"] + np.mean(out["logp"] / 105)) < 1e-14


def test_rbf_regression_vs_sklearn_predict():
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import ConstantKernel as C, RBF, WhiteKernel
    rng = np.random.default_rng(0)
    z = np.abs(rng.standard_normal((19, 40))) + 0.3
    y = rng.standard_normal(19)
    hyp = O.GPHypers(np.array([0.8]), np.array([0.1]), np.array([0.3]), lengthscale=2.5)
    sup = [0, 4, 7, 11, 18]
    pred = O.regression_predict(z[sup], y[sup], z, hyp)
    k = C(0.8, "fixed") * RBF(2.5, "fixed") + WhiteKernel(0.3, "fixed")
    gp = GaussianProcessRegressor(kernel=k, alpha=0.0, optimizer=None).fit(z[sup], y[sup] - 0.1)
    mu, sd = gp.predict(z, return_std=True)
    np.testing.assert_allclose(pred["mean"], mu + 0.1, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(pred["var"], sd ** 2, rtol=1e-7, atol=1e-10)   # sklearn's std includes the WhiteKernel
    out = O.regression_episode(z, y, hyp)
    gp_all = GaussianProcessRegressor(kernel=k, alpha=0.0, optimizer=None).fit(z, y - 0.1)
    assert abs(gp_all.log_marginal_likelihood_value_ - out["logp"][0]) < 1e-9 * abs(out["logp"][0])


def test_closed_form_grads_vs_autograd_linear_and_rbf():
    z = O.synthetic_features(1, 30, 24, 9)[0]
    hyp = O.perturbed_hypers(3, 1)
    out = O.train_episode(z, 3, hyp)
    zt = torch.tensor(z, requires_grad=True)
    s = torch.tensor(hyp.outputscale, requires_grad=True)
    m = torch.tensor(hyp.mean, requires_grad=True)
    nz = torch.tensor(hyp.noise, requires_grad=True)
    loss, logp, alpha = T.classification_loss(zt, 3, s, m, nz)
    loss.backward()
    assert abs(loss.item() - out["loss"]) < 1e-12
    np.testing.assert_allclose(zt.grad.numpy(), out["dz"], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(s.grad.numpy(), out["dsv"], rtol=1e-8)
    np.testing.assert_allclose(m.grad.numpy(), out["dmean"], rtol=1e-8)
    np.testing.assert_allclose(nz.grad.numpy(), out["dnoise"], rtol=1e-8)
    # rbf
    hyp.lengthscale = 0.9
    out = O.train_episode(z, 3, hyp, "rbf")
    zt = torch.tensor(z, requires_grad=True)
    ls = torch.tensor(0.9, dtype=torch.float64, requires_grad=True)
    loss, _, _ = T.classification_loss(zt, 3, torch.tensor(hyp.outputscale), torch.tensor(hyp.mean),
                                       torch.tensor(hyp.noise), "rbf", ls)
    loss.backward()
    np.testing.assert_allclose(zt.grad.numpy(), out["dz"], rtol=1e-7, atol=1e-12)
    assert abs(ls.grad.item() - out["dlengthscale"]) < 1e-8 * abs(out["dlengthscale"])


def test_front_half_matches_torch():
    rng = np.random.default_rng(2)
    zr = rng.standard_normal((21, 16)) * 2 + 1
    bn = torch.nn.BatchNorm1d(16).double()
    bn.train()
    zt = bn(torch.tensor(zr))
    zo, mu, var_u = O.batchnorm1d_train(zr)
    np.testing.assert_allclose(zo, zt.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(bn.running_mean.numpy(), 0.1 * mu, atol=1e-12)
    np.testing.assert_allclose(bn.running_var.numpy(), 0.9 + 0.1 * var_u, atol=1e-12)
    np.testing.assert_allclose(O.l2_normalize(zo), torch.nn.functional.normalize(zt, p=2, dim=1).detach().numpy(), atol=1e-13)
    bn.eval()
    np.testing.assert_allclose(O.batchnorm1d_eval(zr, bn.running_mean.numpy(), bn.running_var.numpy()),
                               bn(torch.tensor(zr)).detach().numpy(), atol=1e-12)


def test_targets_and_classify():
    y = O.one_vs_rest_targets(3, 2)
    assert y.tolist() == [[1, 1, -1, -1, -1, -1], [-1, -1, 1, 1, -1, -1], [-1, -1, -1, -1, 1, 1]]
    mu = np.array([[0.2, 0.5], [0.2, 0.1], [0.1, 0.5]])
    assert O.classify(mu).tolist() == [0, 0]       # ties -> first maximum (np.argmax)


def test_jitter_retry_semantics():
    g = np.load(os.path.join(GOLD, "degenerate.npz"))
    e = O.gram_linear(g["z"])
    y = O.one_vs_rest_targets(2, 6)
    res = O.mll_terms(e, y, np.array([0.7, 1.1]), np.array([0.05, -0.02]), np.array([0.1, 0.1]))
    assert (res.jitter == 0).all()
    np.testing.assert_allclose(res.logp, g["logp_dup"], rtol=1e-12)
    # a matrix whose smallest eigenvalue is -5e-5: 1e-6 and 1e-5 fail, 1e-4 succeeds
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.standard_normal((8, 8)))
    k = q @ np.diag([-5e-5, 0.3, 0.5, 0.7, 1.0, 1.2, 1.5, 2.0]) @ q.T
    l, jit = O.psd_safe_cholesky(k, 1e-6, 3)
    assert jit == pytest.approx(1e-4)
    with pytest.raises(O.NotPSDError):
        O.psd_safe_cholesky(k - 0.01 * np.eye(8), 1e-6, 3)


def test_eval_episode_golden():
    for path in sorted(glob.glob(os.path.join(GOLD, "test_*.npz"))):
        g = np.load(path)
        c, s, q, d = int(g["n_way"]), int(g["n_support"]), int(g["n_query"]), int(g["d"])
        zall = O.synthetic_features(1, c * (s + q), d, int(g["seed"]), c)[0].reshape(c, s + q, d)
        assert abs(zall.sum() - float(g["z_checksum"])) < 1e-9
        out = O.eval_episode(zall[:, :s].reshape(c * s, d), zall[:, s:].reshape(c * q, d), c,
                             O.GPHypers(g["outputscale"], g["mean"], g["noise"]))
        np.testing.assert_allclose(out["mu"], g["mu"], rtol=1e-9, atol=1e-11)
        assert (out["labels"] == g["labels"]).all()


def test_spectral_mixture_restatement_and_golden():
    """SpectralMixtureKernel (DKT_regression.py:121-122): the numpy restatement against (i) the closed form for one
    mixture in one dimension, (ii) an independent torch formulation, (iii) the committed fixture (generated with a scipy
    logpdf cross-check, tests/golden/make_golden.py)."""
    import torch
    from oracle import dkt_oracle_torch as T
    x = np.array([[0.3], [-0.2], [1.1]])
    e = O.gram_spectral_mixture(x, None, [0.7], [[0.4]], [[0.9]])
    tau = x - x.T
    np.testing.assert_allclose(e, 0.7 * np.exp(-2 * np.pi ** 2 * (0.9 * tau) ** 2) * np.cos(2 * np.pi * 0.4 * tau), rtol=1e-13)
    g = np.load(os.path.join(GOLD, "regression_spectral_q4.npz"))
    hyp = O.GPHypers(np.ones(1), g["mean"], g["noise"], mixture=(g["weights"], g["means"], g["scales"]))
    out = O.regression_episode(g["z"], g["labels"], hyp, kernel="spectral")
    np.testing.assert_allclose(out["e"], g["e"], rtol=1e-12, atol=1e-15)
    assert abs(out["loss"] - float(g["loss"])) < 1e-12 and abs(out["logp"][0] - float(g["logp_scipy"])) < 1e-9 * abs(out["logp"][0])
    et = T.spectral_mixture(torch.tensor(g["z"]), None, torch.tensor(g["weights"]), torch.tensor(g["means"]), torch.tensor(g["scales"]))
    np.testing.assert_allclose(et.numpy(), g["e"], rtol=1e-11, atol=1e-14)
    # the matrix is symmetric PSD with sum(w) on the diagonal
    np.testing.assert_allclose(np.diag(out["e"]), np.full(19, g["weights"].sum()), rtol=1e-13)
    assert np.linalg.eigvalsh(out["e"]).min() > -1e-10
    sup = g["support"]
    pred = O.regression_predict(g["z"][sup], g["labels"][sup], g["z"], hyp, kernel="spectral")
    np.testing.assert_allclose(pred["mean"], g["pred_mean"], rtol=1e-10)
    np.testing.assert_allclose(pred["var"], g["pred_var"], rtol=1e-10)


def test_cfg0_fixture_matches_the_oracle_and_its_generator():
    """tests/golden/cfg0_qmul_regression_rbf.npz (BASELINE.json configs[0] shape): the committed values are what the float64
    oracle computes from the regenerated features (scikit-learn / scipy cross-checks are stored next to them)."""
    import os
    import sys
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    from make_golden import cfg0_features
    g = np.load(os.path.join(gold, "cfg0_qmul_regression_rbf.npz"))
    z, labels = cfg0_features(int(g["seed"]), int(g["n"]), int(g["d"]))
    hyp = O.GPHypers(g["outputscale"], g["mean"], g["noise"], lengthscale=float(g["lengthscale"]))
    out = O.regression_episode(z, labels, hyp, kernel="rbf")
    assert abs(out["loss"] - float(g["loss"])) < 1e-12
    assert abs(float(g["logp_sklearn"]) - out["logp"][0]) < 1e-8 * abs(out["logp"][0])
    assert abs(float(g["logp_scipy"]) - out["logp"][0]) < 1e-8 * abs(out["logp"][0])
    np.testing.assert_allclose(out["dz"][:3], g["dz_rows"], rtol=1e-10, atol=1e-14)
    pred = O.regression_predict(z[g["support"]], labels[g["support"]], z, hyp, kernel="rbf")
    np.testing.assert_allclose(pred["mean"], g["pred_mean"], rtol=1e-10)
    np.testing.assert_allclose(pred["var"], g["pred_var"], rtol=1e-10)
