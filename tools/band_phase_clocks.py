"""Per-phase shader-clock breakdown of the two two-sided kernels of the band marginal-likelihood path (a -DDKT_BAND_CLOCKS build of the library: thread 0 of every
workgroup accumulates s_memtime ticks per phase and reports them through alpha[b, 0 / 1, 0..7]).  Measurement tooling.
    python tools/band_phase_clocks.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402

var = os.path.join(ROOT, "deep-kernel-transfer_amd", "libdkt_bclk.so")      # prebuilt in the build container, travels with gpurun
os.environ.setdefault("DKT_EXTRA_HIPCC_FLAGS", "-DDKT_BAND_CLOCKS")
(dkt_amd._lib.build(out=var) if not os.path.exists(var) else None)
os.environ["DKT_AMD_LIB"] = var
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
names = ["V' / R stores + barrier", "pass over the tiles", "partials meet (4 rounds)", "V' -> LDS, T'", "U update, S partial", "Y'", "quadratic form from the residual (back)", "column update", "panel load", "QR (16 columns)", "W store (back)", "alpha store (back)"]
for (b, c, n, d) in [(1024, 20, 420, 128), (256, 20, 420, 128), (64, 20, 420, 128), (1024, 20, 320, 128)]:
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
    al = o["alpha"].double()
    print("B=%d C=%d N=%d: thread 0 of a workgroup, mean s_memtime ticks per phase (100 MHz counter: 1 tick = 10 ns)" % (b, c, n))
    for which, row in (("forward (reduction)", 0), ("back (similarity transform)", 1)):
        v = al[:, row, :12].mean(0)
        tot = v.sum().item()
        print("  %s: total %.0f ticks = %.1f us" % (which, tot, tot / 100.0))
        for nm, x in zip(names, v.tolist()):
            if x:
                print("     %-28s %9.0f  (%4.1f %%)" % (nm, x, 100 * x / tot))
    sys.stdout.flush()
