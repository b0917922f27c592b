"""The backbones adjacent to the hot path against vectors produced by the REFERENCE's own backbone.py
(tests/golden/make_backbone_golden.py): same state-dict keys, same forward in train and eval mode, same
BatchNorm running-statistics update, same gradient -- a reference checkpoint's 'feature.*' tensors drop in unchanged."""
import os

import numpy as np
import pytest
import torch

import dkt_amd
from backbone_fill import CASES, fill_state, make_input, run_case

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "backbone_reference.npz"))


@pytest.mark.parametrize("idx", range(len(CASES)), ids=[c[0] for c in CASES])
def test_backbone_matches_reference_vectors(idx):
    name, size, batch, ch = CASES[idx]
    torch.manual_seed(0)
    torch.set_num_threads(4)
    m = getattr(dkt_amd.backbone, name)()
    assert sum(p.numel() for p in m.parameters()) == int(GOLD[name + "/n_params"])
    fill_state(m, seed=10 + idx)
    res = run_case(m, make_input(size, batch, ch, idx))
    assert list(res["shape"]) == list(GOLD[name + "/shape"])
    for key in ("train_out_head", "eval_out_head", "running_mean_last"):
        np.testing.assert_allclose(res[key], GOLD[name + "/" + key], rtol=2e-4, atol=2e-6, err_msg=name + "/" + key)
    for key in ("train_out_sum", "train_out_abs", "eval_out_sum", "loss", "grad_first_conv_sum", "grad_first_conv_abs"):
        ref = float(GOLD[name + "/" + key])
        scale = max(abs(ref), float(GOLD[name + "/train_out_abs"]) * 1e-3 if "out" in key else abs(ref), 1e-12)
        assert abs(float(res[key]) - ref) <= 5e-4 * scale, (name, key, float(res[key]), ref)
