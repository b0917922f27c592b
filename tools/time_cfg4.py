"""cfg4-class shapes (20-way: N = 320 / 420, D = 512): time of the generic large-N path."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
for (c, per, d, b) in [(20, 16, 512, 64), (20, 21, 512, 64), (5, 38, 512, 256)]:
    n = c * per
    z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev), dim=2).contiguous().requires_grad_(True)
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.7, device=dev)
    mean = torch.zeros(c, device=dev)
    noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)

    def step():
        z.grad = None
        obj = ops.episode_loss_linear(z, y, sv, mean, noise, cw)[0]
        obj.mean().backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ops.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    kt = ops.kernel_timing_results()
    ops.kernel_timing(False)
    print("N=%d C=%d D=%d B=%d: %.2f ms/step = %.0f episodes/s  kernels %s" % (n, c, d, b, 1e3 * dt, b / dt, {k: round(v[1], 3) for k, v in kt.items()}), flush=True)
