"""Time the tile-array marginal-likelihood kernels (N > 127) at the 20-way shapes; run under rocprofv3 --kernel-trace --stats for the
per-kernel split.  Measurement tooling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
shapes = [(1024, 20, 420, 128), (1024, 20, 320, 128)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1:]]
for (b, c, n, d) in shapes:
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    if cls.numel() < n:
        cls = torch.cat([cls, torch.zeros(n - cls.numel(), dtype=cls.dtype, device=dev)])
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.7, device=dev); mean = torch.zeros(c, device=dev); noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    e = ops.gram(z)
    flops = c * (n ** 3 / 3 + n ** 3 + 2 * n * n) * b
    for _ in range(2):
        ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        o = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
    t.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(t) / 5
    print("B=%4d C=%2d N=%3d   tiled %.3f ms (%.1f TF)  info %d" % (b, c, n, ms, flops / ms / 1e9, int(o["info"].abs().max())), flush=True)
