"""Per-phase shader-clock breakdown of the tile-array factorisation kernel (a -DDKT_TILED_CLOCKS build of the library: only
tiled_factor_kernel runs and wave 0 of every workgroup reports its clocks through the output arrays).  Measurement tooling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402

var = os.path.join(ROOT, "deep-kernel-transfer_amd", "libdkt_tclk.so")      # prebuilt in the build container, travels with gpurun
os.environ.setdefault("DKT_EXTRA_HIPCC_FLAGS", "-DDKT_TILED_CLOCKS")
(dkt_amd._lib.build(out=var) if not os.path.exists(var) else None)
os.environ["DKT_AMD_LIB"] = var
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
for (b, c, n, d) in [(1024, 20, 420, 128), (1024, 20, 320, 128), (13, 20, 420, 128)]:
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
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    o = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
    t.record()
    torch.cuda.synchronize()
    names = ["form E", "K loop", "own sweeps", "wait for a sweep (barrier)", "panel + barrier + updates", "stores / rest"]
    vals = [o["logp"], o["dsv"], o["dmean"], o["dnoise"], o["jitter"], o["alpha"][:, :, 0]]
    tot = sum(v.double().mean().item() for v in vals)
    print("B=%d C=%d N=%d: factor kernel %.3f ms; wave 0 of a workgroup, mean s_memtime ticks per phase (the counter runs at the shader clock here: ticks = cycles):" % (b, c, n, s.elapsed_time(t)))
    for nm, v in zip(names, vals):
        print("   %-32s %9.0f  (%4.1f %%)" % (nm, v.double().mean().item(), 100 * v.double().mean().item() / tot))
    print("   %-32s %9.0f ticks ~ %.1f us at 2.4 GHz" % ("total", tot, tot / 2400.0), flush=True)
