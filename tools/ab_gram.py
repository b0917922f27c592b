"""A/B of the episode-resident Gram kernels (exact-fp32 MFMA vs 3-way bf16 split): error against float64 and
HIP-event time per launch.  Measurement tooling; prints, never asserts."""
import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=20):
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


for (b, n, d, kind) in [(2048, 105, 1600, "unit"), (2048, 105, 1600, "randn"), (2048, 105, 1600, "wide"), (4096, 75, 512, "randn"),
                        (1024, 128, 1600, "randn"), (2048, 105, 64, "unit"), (512, 105, 2916, "wide"), (256, 100, 36, "randn")]:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.randn(b, n, d, generator=g, device=dev)
    if kind == "unit":
        z = torch.nn.functional.normalize(z, dim=2)
    elif kind == "wide":     # wide dynamic range: exercises the split's low pieces
        z = z * torch.exp(4.0 * torch.randn(b, n, d, generator=g, device=dev))
    z = z.contiguous()
    ref = torch.einsum("bnd,bmd->bnm", z[:8].double(), z[:8].double())
    # condition-aware scale: sum_k |z_ik||z_jk| is what fp32 rounding errors scale with
    mag = torch.einsum("bnd,bmd->bnm", z[:8].double().abs(), z[:8].double().abs())
    line = "%-28s" % str((b, n, d, kind))
    outs = {}
    for split in (0, 1):
        os.environ["DKT_GRAM_SPLIT"] = str(split)
        e = ops.gram(z)
        ms = timed(lambda: ops.gram(z))
        err = ((e[:8].double() - ref).abs() / mag).max().item()
        sym = bool(torch.equal(e, e.transpose(1, 2)))
        outs[split] = e
        gb = 4.0 * b * (n * d + n * n) / 1e9
        line += "  split=%d: %.4f ms %.0f GB/s err/mag %.2e sym %s |" % (split, ms, gb / ms * 1e3, err, sym)
    line += " max|d|/mag %.2e" % ((outs[0][:8].double() - outs[1][:8].double()).abs() / mag).max().item()
    print(line, flush=True)
os.environ.pop("DKT_GRAM_SPLIT", None)

print("---- forward variants <LDS buffers><prefetch depth> at cfg2", flush=True)
z = torch.nn.functional.normalize(torch.randn(2048, 105, 1600, device=dev), dim=2).contiguous()
for rep in range(2):
    for var in ("11", "12", "611", "612"):
        os.environ["DKT_GRAM_SPLIT_VAR"] = var
        ms = timed(lambda: ops.gram(z), reps=50)
        print("var %s: %.4f ms  %.0f GB/s" % (var, ms, 4.0 * 2048 * (105 * 1600 + 105 * 105) / ms / 1e6), flush=True)
os.environ.pop("DKT_GRAM_SPLIT_VAR", None)

print("---- gram_bwd: dZ = s (W + W^T) Z", flush=True)
for (b, n, d) in [(2048, 105, 1600), (4096, 75, 512), (1024, 128, 1600), (1024, 80, 64), (512, 96, 2916), (256, 100, 36)]:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.randn(b, n, d, generator=g, device=dev)
    w = torch.randn(b, n, n, generator=g, device=dev) * torch.exp(2.0 * torch.randn(b, n, n, generator=g, device=dev))
    sc = torch.rand(b, generator=g, device=dev) + 0.5
    ws = (w[:8] + w[:8].transpose(1, 2)).double() * sc[:8].double().view(-1, 1, 1)
    ref = ws @ z[:8].double()
    mag = ws.abs() @ z[:8].double().abs()
    line = "%-20s" % str((b, n, d))
    outs = {}
    for split in (0, 1):
        os.environ["DKT_GRAM_SPLIT"] = str(split)
        dz = ops.gram_bwd(w, z, sc)
        ms = timed(lambda: ops.gram_bwd(w, z, sc))
        err = ((dz[:8].double() - ref).abs() / mag).max().item()
        outs[split] = dz
        gb = 4.0 * b * (2 * n * d + n * n) / 1e9
        line += "  split=%d: %.4f ms %.0f GB/s err/mag %.2e |" % (split, ms, gb / ms * 1e3, err)
    line += " finite %s" % bool(torch.isfinite(outs[1]).all())
    print(line, flush=True)
os.environ.pop("DKT_GRAM_SPLIT", None)

print("---- backward variants <LDS buffers><prefetch depth> at cfg2", flush=True)
z = torch.nn.functional.normalize(torch.randn(2048, 105, 1600, device=dev), dim=2).contiguous()
w = torch.randn(2048, 105, 105, device=dev)
ref = ops.gram_bwd(w, z, None)
for rep in range(2):
    for var in ("22", "11", "12"):
        os.environ["DKT_GRAM_BWD_SPLIT_VAR"] = var
        out = ops.gram_bwd(w, z, None)
        ms = timed(lambda: ops.gram_bwd(w, z, None), reps=50)
        print("var %s: %.4f ms  %.0f GB/s  max|diff| vs var 22 %.2e" % (var, ms, 4.0 * 2048 * (2 * 105 * 1600 + 105 * 105) / ms / 1e6,
                                                                          (out - ref).abs().max().item()), flush=True)
os.environ.pop("DKT_GRAM_BWD_SPLIT_VAR", None)
