"""The lane-level numpy model of the MFMA marginal-likelihood kernels (tools/mll_mfma_model.py) -- the exact register layouts of
v_mfma_f32_16x16x4_f32, the replicated-column diagonal-tile sweep, the three phases in place, and the left-looking tile-array variant
of the large-N path -- against numpy's Cholesky / inverse.  It is the executable statement of the index logic of
csrc/dkt_mll_mfma.hip and csrc/dkt_mll_tiled.hip and runs without a GPU."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("mll_mfma_model", os.path.join(ROOT, "tools", "mll_mfma_model.py"))
model = importlib.util.module_from_spec(spec)
spec.loader.exec_module(model)


def _problem(n, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, 24))
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    k = (0.7 * z @ z.T + 0.1 * np.eye(n)) * scale
    return k, rng.standard_normal(n)


def _check(res, k, r):
    logdet, quad, alpha, p = res
    ki = np.linalg.inv(k)
    a = ki @ r
    assert abs(logdet - np.linalg.slogdet(k)[1]) < 1e-9
    assert abs(quad - r @ a) < 1e-9
    assert np.abs(alpha - a).max() < 1e-9
    assert np.abs(p - (ki - np.outer(a, a))).max() < 1e-9


@pytest.mark.parametrize("n,scale", [(5, 1.0), (15, 37.0), (16, 1.0), (31, 1.0), (47, 5.0)])
def test_wave_per_matrix_model(n, scale):
    k, r = _problem(n, n, scale)
    _check(model.run(n, k, r), k, r)


@pytest.mark.parametrize("n,scale", [(40, 1.0), (79, 5.0)])
def test_tile_array_left_looking_model(n, scale):
    k, r = _problem(n, 100 + n, scale)
    _check(model.run_tiled(n, k, r), k, r)
