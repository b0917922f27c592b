"""Time dkt_mll_f32: the default MFMA wave-per-matrix kernel next to the register-sweep twin (DKT_MLL_FORCE_REG), at the
benchmark shapes.  Measurement tooling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
shapes = [(8192, 5, 105, 64, True), (8192, 5, 85, 64, True), (8192, 5, 25, 64, False), (8192, 5, 25, 64, True), (2048, 5, 105, 64, True),
          (4096, 1, 19, 64, True), (1024, 20, 100, 64, False)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (b, c, n, d, grad) in shapes:
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    if cls.numel() < n:
        cls = torch.cat([cls, torch.zeros(n - cls.numel(), dtype=cls.dtype, device=dev)])
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.7, device=dev) + 0.02 * torch.arange(c, device=dev)
    mean = torch.zeros(c, device=dev)
    noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    e = ops.gram(z)
    res = {}
    for reg in (False, True):
        for _ in range(3):
            out = ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=grad, force_reg=reg)
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            out = ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=grad, force_reg=reg)
        t.record()
        torch.cuda.synchronize()
        res[reg] = (s.elapsed_time(t) / 20, out)
    a, r = res[False][1], res[True][1]
    err = ((a["logp"] - r["logp"]).abs() / r["logp"].abs()).max().item()
    werr = ((a["w"] - r["w"]).norm() / r["w"].norm()).item() if grad else 0.0
    flops = c * (n ** 3 / 3 + (n ** 3 if grad else n ** 3 / 3) + 2 * n * n) * b
    print("B=%5d C=%2d N=%3d grad=%d   mfma %.4f ms (%.1f TF)   reg %.4f ms   logp rel diff %.2e  W rel diff %.2e  info %d" % (
        b, c, n, grad, res[False][0], flops / res[False][0] / 1e9, res[True][0], err, werr, int(a["info"].abs().max())), flush=True)
