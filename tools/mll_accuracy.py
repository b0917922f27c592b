"""Accuracy of dkt_mll_f32 (register kernel vs generic twin) against the float64 oracle at cfg2. Prints, never asserts."""
import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402
from oracle import dkt_oracle as O  # noqa: E402

dev = torch.device("cuda", 0)
c, per, d = 5, 21, 1600
n = c * per
B = 64
rng = np.random.default_rng(0)
z = rng.standard_normal((B, n, d))
z = (z - z.mean(1, keepdims=True)) / np.sqrt(z.var(1, keepdims=True) + 1e-5)
z /= np.linalg.norm(z, axis=2, keepdims=True)
zt = torch.as_tensor(z, dtype=torch.float32, device=dev)
hyp = O.perturbed_hypers(c, 6)
y = O.one_vs_rest_targets(c, per)
cw = np.full(c, -1.0 / (c * n))
t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev)  # noqa: E731
for split in ("1", "0"):
    os.environ["DKT_GRAM_SPLIT"] = split
    e = ops.gram(zt)
    e64 = np.einsum("bnd,bmd->bnm", zt.double().cpu().numpy(), zt.double().cpu().numpy())
    print("gram split=%s  max|E - E64| = %.3e" % (split, np.abs(e.cpu().numpy() - e64).max()))
    for fg in (False, True):
        out = ops.mll(e, t(y), t(hyp.outputscale), t(hyp.mean), t(hyp.noise), want_grad=True, cls_weight=t(cw), force_generic=fg)
        rel_lp, rel_al, rel_w, rel_dsv = 0.0, 0.0, 0.0, 0.0
        for b in range(8):
            ref = O.mll_terms(e64[b], y, hyp.outputscale, hyp.mean, hyp.noise)
            lp = out["logp"][b].cpu().numpy()
            rel_lp = max(rel_lp, float(np.abs((lp - ref.logp) / ref.logp).max()))
            al = out["alpha"][b].cpu().numpy()
            rel_al = max(rel_al, float(np.linalg.norm(al - ref.alpha) / np.linalg.norm(ref.alpha)))
            w_e, _, _, _ = O.mll_grads(e64[b], ref, hyp.outputscale, hyp.noise, cw)
            rel_w = max(rel_w, float(np.linalg.norm(out["w"][b].cpu().numpy() - w_e) / np.linalg.norm(w_e)))
        print("   %-8s logp rel %.3e   alpha relL2 %.3e   W relL2 %.3e" % ("generic" if fg else "register", rel_lp, rel_al, rel_w))
