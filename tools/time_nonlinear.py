"""Training episode with a NON-linear kernel (rbf, per-class lengthscales: one ExactGPLayer per class, reference DKT.py:63-66, 352-365)
through the one-launch path -- one squared-distance Gram per episode, the per-class map element-wise, ONE marginal-likelihood launch over
all (episode, class) matrices (DKT_MLL_E_PER_CLASS) -- against the per-class loop it replaces (C Gram + C single-model launches) and
against the linear (bncossim) episode of the same shape.  Measurement tooling."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
c, per, d = 5, 21, 1600
n = c * per
g = torch.Generator(device=dev).manual_seed(3)
cls = torch.arange(c, device=dev).repeat_interleave(per)
y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.full((c,), 0.69, device=dev)
mean = torch.zeros(c, device=dev)
noise = torch.full((c,), 0.1, device=dev)
cw = torch.full((c,), -1.0 / (c * n), device=dev)
ls = torch.linspace(0.9, 1.4, c, device=dev).requires_grad_(True)


def timed(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


for b in (1, 64, 1024):
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).requires_grad_(True)

    def one_launch_torch_maps():            # round 3 first version: the class maps and their chain rule as torch element-wise ops
        z.grad = None
        e = ops.base_matrix_per_class(z, "rbf", ls)
        obj, *_ = ops.mll_objective(e, y, sv, mean, noise, cw)
        obj.mean().backward()

    def one_launch():                       # the product path: dkt_gram_f32 (SQDIST) -> dkt_class_kernel_f32 -> dkt_mll_f32 -> dkt_class_kernel_bwd_f32 -> dkt_gram_bwd_f32
        z.grad = None
        ls.grad = None
        obj, *_ = ops.episode_loss_class_kernel(z, y, sv, mean, noise, cw, "rbf", ls)
        obj.mean().backward()

    def per_class_loop():
        z.grad = None
        tot = 0.0
        for k in range(c):
            e = ops.base_matrix(z, "rbf", ls[k:k + 1])
            o, *_ = ops.mll_objective(e, y[k:k + 1].contiguous(), sv[k:k + 1], mean[k:k + 1], noise[k:k + 1], cw[k:k + 1])
            tot = tot + o
        tot.mean().backward()

    def linear():
        z.grad = None
        obj, *_ = ops.episode_loss_linear(z, y, sv, mean, noise, cw, unit_rows=True)
        obj.mean().backward()

    a, a0, bb, l = timed(one_launch), timed(one_launch_torch_maps), timed(per_class_loop), timed(linear)
    ops.kernel_timing(True)
    for _ in range(5):
        one_launch()
    torch.cuda.synchronize()
    kt = {k: round(v[1], 4) for k, v in ops.kernel_timing_results().items()}
    ops.kernel_timing(False)
    print("B=%5d N=%d D=%d C=%d rbf, per-class lengthscales: one launch %.3f ms (torch class maps: %.3f) | per-class loop %.3f ms | linear episode %.3f ms | one-launch / linear %.2f\n        kernels: %s" % (
        b, n, d, c, a, a0, bb, l, a / l, kt), flush=True)
