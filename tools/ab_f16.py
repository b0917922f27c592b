"""A/B of the split variants of the episode-resident Gram kernels on unit-norm rows (cfg2, B = 8192): 3-way bf16 split
vs 2-way scaled-f16 split; error against float64 and HIP-event time per launch.  Measurement tooling; prints only."""
import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


variants = sys.argv[1:] or ["11", "211", "212", "2611"]
for (b, n, d) in [(8192, 105, 1600), (8192, 105, 64), (8192, 85, 512)]:
    z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev), dim=2).contiguous()
    ref = torch.einsum("bnd,bmd->bnm", z[:16].double(), z[:16].double())
    for rep in range(2):
        for var in variants:
            os.environ["DKT_GRAM_SPLIT_VAR"] = var
            e = ops.gram(z)
            ms = timed(lambda: ops.gram(z))
            err = (e[:16].double() - ref).abs().max().item()
            sym = bool(torch.equal(e, e.transpose(1, 2)))
            print("%-18s var %-5s %.4f ms  %.0f GB/s  max|E-E64| %.2e  diag err %.2e sym %s" % (
                str((b, n, d)), var, ms, 4.0 * b * (n * d + n * n) / ms / 1e6, err,
                (torch.diagonal(e[:16], dim1=1, dim2=2).double() - 1).abs().max().item(), sym), flush=True)
os.environ.pop("DKT_GRAM_SPLIT_VAR", None)

print("---- gram_bwd: dZ = s (W + W^T) Z, unit-norm rows of Z", flush=True)
bvars = os.environ.get("AB_BWD_VARS", "11 12 222 221 212 211").split()
for (b, n, d) in [(8192, 105, 1600), (8192, 85, 512)]:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    # gradient-like W with a wide per-row dynamic range
    w = torch.randn(b, n, n, generator=g, device=dev) * torch.exp(2.0 * torch.randn(b, n, 1, generator=g, device=dev))
    sc = torch.rand(b, generator=g, device=dev) + 0.5
    ws = (w[:8] + w[:8].transpose(1, 2)).double() * sc[:8].double().view(-1, 1, 1)
    ref = ws @ z[:8].double()
    mag = ws.abs() @ z[:8].double().abs()
    for rep in range(2):
        for var in bvars:
            os.environ["DKT_GRAM_BWD_SPLIT_VAR"] = var
            dz = ops.gram_bwd(w, z, sc)
            ms = timed(lambda: ops.gram_bwd(w, z, sc))
            err = ((dz[:8].double() - ref).abs() / mag).max().item()
            rel = ((dz[:8].double() - ref).norm() / ref.norm()).item()
            print("%-18s var %-4s %.4f ms  %.0f GB/s  max err/mag %.2e  rel-L2 %.2e finite %s" % (
                str((b, n, d)), var, ms, 4.0 * b * (2 * n * d + n * n) / ms / 1e6, err, rel, bool(torch.isfinite(dz).all())), flush=True)
os.environ.pop("DKT_GRAM_BWD_SPLIT_VAR", None)
