"""Occupancy A/B of the band path's kernels (dkt_mll_band.hip; twins library): DKT_PAD_BAND_FWD / _BACK / _CLASS bytes of untouched dynamic LDS per launch (one workgroup
per CU instead of two for the two-sided kernels; 8 / 4 / 2 workgroups for the class kernel).  1024 episodes of N = 420 / 320, C = 20.   python tools/band_occ_ab.py"""
import os
import sys

os.environ["DKT_TWINS"] = "force"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


for (b, c, n, d) in [(1024, 20, 420, 512), (1024, 20, 320, 512)]:
    g = torch.Generator(device=dev).manual_seed(n)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2)
    e = (z @ z.transpose(1, 2)).contiguous()
    cls = torch.arange(c, device=dev).repeat_interleave(n // c)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.linspace(0.8, 1.4, c, device=dev)
    mean = torch.zeros(c, device=dev)
    noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    fn = lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, force_band=True)
    base = timed(fn)
    line = "B=%d C=%d N=%d  default %.3f ms" % (b, c, n, base)
    for name, pads in (("DKT_PAD_BAND_BACK", ("30000",)), ("DKT_PAD_BAND_FWD", ("30000",)), ("DKT_PAD_BAND_CLASS", ("20000", "40000", "80000"))):
        for p in pads:
            os.environ[name] = p
            line += "   %s=%s %.3f" % (name[8:], p, timed(fn))
        del os.environ[name]
    line += "   default again %.3f" % timed(fn)
    print(line, flush=True)
