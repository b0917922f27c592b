"""Level-1 training episode (ops.episode_loss_linear fwd + bwd) at small batches: the feature-space path (5 launches + [C]-sized glue) against the N x N path
(3 launches) at the Omniglot 5-way shape -- where does the launch count stop mattering?  Measurement tooling; prints, never asserts."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
for (c, per, d) in ((5, 21, 64), (20, 21, 64)):
    n = c * per
    for b in (1, 8, 64, 256, 1024):
        z = bench.synthetic_batch(b, n, d, 3, dev).requires_grad_(True)
        y = torch.where(torch.arange(c, device=dev).repeat_interleave(per).unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
        raw_s, mean = bench.perturbed_hypers(c, 99, dev)
        raw_s.requires_grad_(True)
        mean.requires_grad_(True)
        noise = torch.full((c,), 0.1, device=dev)
        cw = torch.full((c,), -1.0 / (c * n), device=dev)
        res = []
        for lr in ("1", "0"):
            os.environ["DKT_LOWRANK"] = lr

            def step():
                z.grad = None
                out = ops.episode_loss_linear(z, y, torch.nn.functional.softplus(raw_s), mean, noise, cw, unit_rows=True)
                out[0].mean().backward()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                step()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 30 * 1e3)
        print("C=%d N=%d D=%d B=%5d  feature-space %.3f ms   N x N %.3f ms" % (c, n, d, b, res[0], res[1]))
