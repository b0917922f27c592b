"""Race hunt: every batched kernel at full occupancy, many launches on the same inputs, bitwise comparison."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for (c, per, d) in [(5, 21, 1600), (5, 15, 512), (4, 24, 256), (2, 64, 128), (5, 17, 1024), (1, 127, 64), (3, 37, 64)]:
    n, b = c * per, 2048
    g = torch.Generator(device=dev).manual_seed(n + d)
    x = torch.randn(b, n, d, generator=g, device=dev).abs() + 0.3
    gamma = torch.rand(d, generator=g, device=dev) + 0.5
    beta = torch.randn(d, generator=g, device=dev) * 0.1
    st = ops.bn_stats(x, gamma, beta)
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.rand(c, generator=g, device=dev) + 0.3
    mean = torch.randn(c, generator=g, device=dev) * 0.1
    noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    sc = torch.rand(b, generator=g, device=dev) + 0.5
    ref = None
    for r in range(reps):
        e, rn = ops.gram_bn(x, st["a"], st["s"])
        zn = ((x * st["a"].unsqueeze(1) + st["s"].unsqueeze(1)) * rn.unsqueeze(2)).contiguous()
        e2 = ops.gram(zn)
        out = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
        dz = ops.gram_bwd(out["w"], zn, sc)
        dx, dg, db = ops.gram_bn_bwd(out["w"], e, x, st["a"], st["s"], rn, st["mean"], st["rstd"], sc)
        cur = [e, rn, e2, out["logp"], out["alpha"], out["w"], out["dsv"], dz, dx, dg, db]
        if ref is None:
            ref = [t.clone() for t in cur]
        else:
            for k, (a, bb) in enumerate(zip(ref, cur)):
                if not torch.equal(a, bb):
                    bad += 1
                    print("MISMATCH shape", (c, per, d), "rep", r, "tensor", k, "episodes", int(((a - bb).flatten(1).abs().max(1).values > 0).sum()), flush=True)
    print("shape", (c, per, d), "N", n, "ok so far, mismatches:", bad, flush=True)
# large-N shapes (tile-array marginal-likelihood kernels, row-block Gram kernels) and the unit-row paths of the headline shape
for (c, per, d, b, unit) in [(20, 21, 128, 256, True), (20, 16, 128, 256, True), (5, 38, 64, 512, True), (5, 21, 1600, 2048, True), (2, 150, 96, 256, False)]:
    n = c * per
    g = torch.Generator(device=dev).manual_seed(n + d + 1)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.rand(c, generator=g, device=dev) + 0.3
    mean = torch.randn(c, generator=g, device=dev) * 0.1
    noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    sc = torch.rand(b, generator=g, device=dev) + 0.5
    ref = None
    for r in range(reps):
        e = ops.gram(z, kind=ops.KERNEL_LINEAR_UNIT if unit else ops.KERNEL_LINEAR)
        out = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
        dz = ops.gram_bwd(out["w"], z, sc, unit_rows=unit, w_symmetric=unit)
        cur = [e, out["logp"], out["alpha"], out["w"], out["dsv"], out["dmean"], out["dnoise"], dz]
        if ref is None:
            ref = [t.clone() for t in cur]
        else:
            for k, (a, bb) in enumerate(zip(ref, cur)):
                if not torch.equal(a, bb):
                    bad += 1
                    print("MISMATCH shape", (c, per, d, b), "rep", r, "tensor", k, flush=True)
    print("shape", (c, per, d, b, unit), "N", n, "ok so far, mismatches:", bad, flush=True)
print("TOTAL MISMATCHES", bad)
