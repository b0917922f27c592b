"""The block-level numpy model of csrc/dkt_mll_band.hip (tools/band_mll_model.py: one orthogonal reduction of E to block-tridiagonal form per episode, every class a
block LDL^T of B + mu_c I, one similarity transform back) against the float64 oracle.  It is the executable statement of the kernels' algorithm -- the same loops,
operand orientations and recurrences -- and runs without a GPU."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import dkt_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("band_mll_model", os.path.join(ROOT, "tools", "band_mll_model.py"))
model = importlib.util.module_from_spec(spec)
spec.loader.exec_module(model)


def _rel(a, r):
    return float(np.linalg.norm(np.asarray(a) - r) / np.linalg.norm(r))


@pytest.mark.parametrize("n,d,c,corr,dtype,tol", [(42, 24, 3, 0, np.float64, 1e-12), (150, 64, 10, 10, np.float64, 1e-11), (130, 48, 5, 0, np.float32, 2e-5)])
def test_band_model_matches_oracle(n, d, c, corr, dtype, tol):
    z = O.synthetic_features(1, n, d, 5, corr)[0]
    hyp = O.perturbed_hypers(c, 9)
    y = O.one_vs_rest_targets(c, n // c)
    cw = np.full(c, -1.0 / (c * n))
    e = z @ z.T
    res = O.mll_terms(e, y, hyp.outputscale, hyp.mean, hyp.noise)
    w_ref, _, _, _ = O.mll_grads(e, res, hyp.outputscale, hyp.noise, cw)
    _, dsv, dmean, dnoise = O.mll_grads(e, res, hyp.outputscale, hyp.noise, np.ones(c))
    out = model.episode(e.astype(dtype), y, hyp.outputscale, hyp.mean, hyp.noise, cw, dtype)
    assert not out["info"].any()
    assert np.abs((out["logp"] - res.logp) / res.logp).max() < tol
    assert _rel(out["alpha"], res.alpha) < 10 * tol
    assert _rel(out["w"], w_ref) < 10 * tol
    for got, ref in ((out["dsv"], dsv), (out["dmean"], dmean), (out["dnoise"], dnoise)):
        assert _rel(got, ref) < 30 * tol
    band = out["band"]
    i, j = np.indices(band.shape)
    assert np.abs(band[np.abs(i // 16 - j // 16) > 1]).max() == 0.0            # block tridiagonal


def test_band_model_flags_a_singular_class():
    """Rank-deficient E with zero noise: the block LDL^T meets a non-positive pivot (attempt 0 of psd_safe_cholesky fails; the kernels hand the episode to the ladder)."""
    z = O.synthetic_features(1, 60, 16, 3, 0)[0]
    y = O.one_vs_rest_targets(3, 20)
    out = model.episode(z @ z.T, y, np.ones(3), np.zeros(3), np.array([0.0, 0.1, 0.0]), np.full(3, -1.0 / 180), np.float32)
    assert out["info"][1] == 0 and out["info"][0] > 0 and out["info"][2] > 0
