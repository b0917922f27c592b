"""dkt_mll_f32 (training call, N <= 127) over the batch size: the wave-per-episode kernel (DKT_MLL_H2E_MINB=1) against the
wave-per-matrix kernel (DKT_MLL_H2E_MINB=1000000000), both on the f16 matrix pipe.  Run once per setting of the variable:
    DKT_MLL_H2E_MINB=1 python tools/time_mll_batch.py        Measurement tooling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
print("DKT_MLL_H2E_MINB =", os.environ.get("DKT_MLL_H2E_MINB"))
for (c, n, d) in ((5, 105, 64), (5, 85, 64), (5, 65, 64)):
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.69, device=dev) * (1.0 + 0.03 * torch.arange(c, device=dev))
    mean = 0.01 * torch.arange(c, device=dev, dtype=torch.float32)
    noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    z = torch.nn.functional.normalize(torch.randn(2048, n, d, generator=g, device=dev), dim=2).contiguous()
    e0 = ops.gram(z)
    row = []
    for b in (256, 512, 1024, 2048, 4096, 8192, 16384):
        e = e0.repeat((b + 2047) // 2048, 1, 1)[:b].contiguous()
        for _ in range(3):
            ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=True)
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=True)
        t.record()
        torch.cuda.synchronize()
        row.append("B=%d: %.4f ms (%.1f us/1k ep)" % (b, s.elapsed_time(t) / 20, s.elapsed_time(t) / 20 / b * 1e6))
    print("C=%d N=%d  " % (c, n) + " | ".join(row), flush=True)
