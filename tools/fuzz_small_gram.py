"""Randomised parity sweep of the N <= 32 Gram kernels (dkt_gram_small.hip: the product's default dispatch, incl. the rows-beyond-sixteen-on-the-VALU instances of 17 .. 20 rows
and the occupancy caps) against float64: random (B, N, D, kind, lengthscale, scale), D any multiple of 4, forward + backward.   python tools/fuzz_small_gram.py [cases]"""
import os
import sys

os.environ["DKT_TWINS"] = "0"
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(2026)
worst = {"fwd": 0.0, "bwd": 0.0, "asym": 0}
bad = 0
for it in range(cases):
    n = int(rng.choice([17, 18, 19, 20, 19, 19, 16, 21, 5, 10, 25, 32]))
    d = 4 * int(rng.integers(1, 1100))
    b = int(rng.choice([1, 2, 3, 7, 64, 300, 1024, 4096, 5000]))
    if b * n * d > 3e8:
        b = max(1, int(3e8 // (n * d)))
    kind = ops.KERNEL_RBF if rng.random() < 0.6 else ops.KERNEL_LINEAR
    scale = float(rng.choice([0.02, 0.05, 0.3, 1.0]))
    ls = float(rng.uniform(0.5, 3.0)) * max(1.0, scale * np.sqrt(d) / 2)
    g = torch.Generator(device=dev).manual_seed(it)
    z = torch.randn(b, n, d, device=dev, generator=g) * scale + (0.5 * scale if it % 3 == 0 else 0.0)
    w = torch.randn(b, n, n, device=dev, generator=g) * 0.1
    lst = torch.tensor([ls], device=dev, dtype=torch.float32)
    e = ops.gram(z, None, kind, lst if kind == ops.KERNEL_RBF else None)
    dz = ops.gram_bwd(w, z)
    k = min(b, 16)
    zd = z[:k].double()
    ref = zd @ zd.transpose(1, 2)
    if kind == ops.KERNEL_RBF:
        dg = torch.diagonal(ref, dim1=1, dim2=2)
        ref = torch.exp(-0.5 * (dg.unsqueeze(2) + dg.unsqueeze(1) - 2 * ref).clamp_min(0) / ls ** 2)
    ef = ((e[:k].double() - ref).abs().max() / ref.abs().max()).item()
    wd = w[:k].double()
    dref = (wd + wd.transpose(1, 2)) @ zd
    eb = ((dz[:k].double() - dref).norm() / dref.norm()).item()
    sym = bool(torch.equal(e, e.transpose(1, 2)))
    # RBF: the distances are differences of O(|z|^2 D) terms: tolerance relative to that cancellation
    tol_f = 4e-6 if kind == ops.KERNEL_LINEAR else 4e-6 * max(1.0, float((zd * zd).sum(2).max()) / ls ** 2)
    worst["fwd"] = max(worst["fwd"], ef / tol_f)
    worst["bwd"] = max(worst["bwd"], eb / 4e-6)
    if not sym:
        worst["asym"] += 1
    if ef > tol_f or eb > 4e-6 or not sym or not torch.isfinite(e).all() or not torch.isfinite(dz).all():
        bad += 1
        print("MISMATCH case %d: B=%d N=%d D=%d kind=%d scale=%g ls=%g  fwd %.2e (tol %.1e) bwd %.2e sym %s" % (it, b, n, d, kind, scale, ls, ef, tol_f, eb, sym), flush=True)
print("%d cases, %d mismatches; worst error / tolerance: forward %.2f, backward %.2f; asymmetric outputs %d" % (cases, bad, worst["fwd"], worst["bwd"], worst["asym"]), flush=True)
