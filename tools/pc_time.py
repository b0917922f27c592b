"""20-way episode of 420 rows with a kernel whose class models own their lengthscale (rbf): the training step's GP part in ONE per-class
dkt_mll_f32 call (tile-array pipeline, DKT_MLL_E_PER_CLASS) against the per-class loop of single-model calls it replaces.  Measurement tooling."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for (b, c, n, d) in [(64, 20, 420, 64), (16, 5, 150, 64), (1, 20, 420, 64)]:
    per = n // c
    z = (0.3 * torch.randn(b, n, d, generator=g, device=dev)).requires_grad_(True)
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.linspace(0.5, 1.5, c, device=dev).requires_grad_(True)
    mean = torch.zeros(c, device=dev, requires_grad=True)
    noise = torch.full((c,), 0.1, device=dev, requires_grad=True)
    ls = torch.linspace(2.0, 4.0, c, device=dev).requires_grad_(True)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)

    def one_launch():
        for p in (z, sv, mean, noise, ls):
            p.grad = None
        obj, logp, alpha, info, jit, e = ops.episode_loss_class_kernel(z, y, sv, mean, noise, cw, "rbf", ls, None)
        obj.mean().backward()
        return obj.detach().clone(), z.grad.clone(), ls.grad.clone(), info

    def loop():
        for p in (z, sv, mean, noise, ls):
            p.grad = None
        objs = []
        for k in range(c):
            e = ops.base_matrix(z, "rbf", ls[k:k + 1])
            o, lp, al, inf, jt = ops.mll_objective(e, y[k:k + 1].contiguous(), sv[k:k + 1], mean[k:k + 1], noise[k:k + 1], cw[k:k + 1])
            objs.append(o)
        obj = torch.stack(objs, 0).sum(0)
        obj.mean().backward()
        return obj.detach().clone(), z.grad.clone(), ls.grad.clone()

    o1, dz1, dl1, info = one_launch()
    o2, dz2, dl2 = loop()
    rel = lambda x, r: ((x - r).norm() / r.norm()).item()
    print("B=%d C=%d N=%d: info max %d; obj rel %.2e, dz rel %.2e, dls rel %.2e (one launch vs per-class loop)" % (b, c, n, int(info.abs().max()), rel(o1, o2), rel(dz1, dz2), rel(dl1, dl2)))
    t1 = timed(one_launch)
    t2 = timed(loop)
    print("   one per-class call %.2f ms, per-class loop %.2f ms  (x%.1f)" % (t1, t2, t2 / t1), flush=True)
