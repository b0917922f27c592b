"""Per-phase shader-clock breakdown of the resident-accumulator W kernel of the tile-array marginal likelihood (a -DDKT_WRES_CLOCKS build:
wave 0 of every workgroup reports its clocks through the per-class output arrays, slot (episode, column range)).  Measurement tooling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402

var = os.path.join(ROOT, "deep-kernel-transfer_amd", "libdkt_wclk.so")      # prebuilt in the build container, travels with gpurun
if not os.path.exists(var):
    os.environ.setdefault("DKT_EXTRA_HIPCC_FLAGS", "-DDKT_WRES_CLOCKS")
    dkt_amd._lib.build(out=var)
os.environ["DKT_AMD_LIB"] = var
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
for (b, c, n, d) in [(1024, 20, 420, 128), (1024, 20, 320, 128)]:
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.7, device=dev); mean = torch.zeros(c, device=dev); noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    e = ops.gram(z)
    for _ in range(2):
        o = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
    torch.cuda.synchronize()
    names = ["products (ds_read + MFMA issue)", "wait for the staged loads", "split + LDS write", "issue next loads + group logic", "barrier"]
    vals = [o["logp"], o["dsv"], o["dmean"], o["dnoise"], o["jitter"]]
    nt = (n + 1 + 15) // 16
    print("B=%d C=%d N=%d (NT = %d): wave 0 of a workgroup, mean s_memtime ticks per phase over the whole (class, strip) stream:" % (b, c, n, nt))
    for gi in range(4):
        col = [v[:, gi].double().mean().item() for v in vals]
        tot = sum(col)
        if tot <= 0 or not all(x >= 0 for x in col) or tot < 1000:
            continue
        print("  column range %d: total %9.0f ticks" % (gi, tot))
        for nm, x in zip(names, col):
            print("     %-34s %9.0f  (%4.1f %%)" % (nm, x, 100 * x / tot))
    sys.stdout.flush()
