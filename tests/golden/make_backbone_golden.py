"""Generates tests/golden/backbone_reference.npz by running the REFERENCE's own backbone.py (importable on CPU in the build
container, SURVEY.md 8c "Backbone oracle") on seeded weights and inputs.  This is the one component adjacent to the hot path
whose reference implementation runs here, so these vectors are pinned against the reference itself.  The reference never
travels: only this script and the small output arrays are committed.   Re-run:  python tests/golden/make_backbone_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
import backbone as ref_backbone  # noqa: E402  (the reference's module)
from backbone_fill import CASES, fill_state, make_input, run_case  # noqa: E402

torch.set_num_threads(4)
out = {}
for i, (name, size, batch, ch) in enumerate(CASES):
    torch.manual_seed(0)
    m = fill_state(getattr(ref_backbone, name)(), seed=10 + i)
    res = run_case(m, make_input(size, batch, ch, i))
    for k, v in res.items():
        out["%s/%s" % (name, k)] = v
    out["%s/n_params" % name] = np.array(sum(p.numel() for p in m.parameters()))
    out["%s/final_feat_dim" % name] = np.array(getattr(m, "final_feat_dim", -1) if not isinstance(getattr(m, "final_feat_dim", -1), list) else -2)
    print(name, res["shape"], float(res["loss"]))
np.savez_compressed(os.path.join(HERE, "backbone_reference.npz"), **out)
print("wrote backbone_reference.npz", os.path.getsize(os.path.join(HERE, "backbone_reference.npz")), "bytes")
