"""Where the band reduction (csrc/dkt_mll_band.hip) beats the tile-array kernels (force_tiled): marginal likelihood with gradients, ms per call, over (B, C, N).
Measurement tooling behind the dispatch rule of dkt_mll_band_supports().    python tools/band_crossover.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)


def timed(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) / reps


print("    B   C    N    band ms   tiled ms   tiled / band")
for (b, c, n) in [(1024, 8, 128), (1024, 8, 256), (1024, 8, 420), (1024, 10, 200), (1024, 10, 320), (1024, 10, 420), (1024, 12, 256), (1024, 16, 192), (1024, 16, 256), (1024, 16, 320),
                  (1024, 20, 200), (1024, 20, 256), (1024, 20, 320), (1024, 20, 420), (1024, 32, 256), (1024, 32, 420),
                  (512, 20, 420), (256, 20, 420), (128, 20, 420), (64, 20, 420), (256, 20, 320), (128, 20, 320), (256, 16, 256), (128, 16, 256)]:
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    if cls.numel() < n:
        cls = torch.cat([cls, torch.zeros(n - cls.numel(), dtype=cls.dtype, device=dev)])
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.7, device=dev); mean = torch.zeros(c, device=dev); noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    z = torch.nn.functional.normalize(torch.randn(b, n, 64, generator=g, device=dev), dim=2).contiguous()
    e = ops.gram(z)
    tb = timed(lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, force_band=True))
    tt = timed(lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, force_tiled=True))
    print("%5d  %2d  %3d   %8.3f   %8.3f   %6.2f" % (b, c, n, tb, tt, tt / tb), flush=True)
