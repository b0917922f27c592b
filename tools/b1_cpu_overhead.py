"""Host time per call of the level-1 step's pieces at B = 1 (the eager loop is bound by it): every ops.* call of the step and the tensor expressions the two reduction
launches replace, 2000 back-to-back calls each, one synchronize at the end.  Measurement tooling."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
b, n, d, c = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 105, 1600, 5
z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev), dim=2)
cls = torch.arange(c, device=dev).repeat_interleave(n // c)
y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv, mean, noise = torch.ones(c, device=dev), torch.zeros(c, device=dev), torch.full((c,), 0.1, device=dev)
cw = torch.full((c,), -1.0 / (c * n), device=dev)
e = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
o = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
gobj = torch.ones(b, device=dev)
shapes = (sv.shape, mean.shape, noise.shape)


def torch_obj():
    return (o["logp"] * cw.reshape(1, -1)).sum(1)


def torch_hyper():
    gw = gobj.reshape(-1, 1) * cw.reshape(1, -1)
    return (gw * o["dsv"]).sum(0).reshape(shapes[0]), (gw * o["dmean"]).sum(0).reshape(shapes[1])


cases = [("ops.gram", lambda: ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)), ("ops.mll", lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)),
         ("ops.gram_bwd", lambda: ops.gram_bwd(o["w"], z, gobj, unit_rows=True, w_symmetric=True)), ("ops.objective", lambda: ops.objective(o["logp"], cw)),
         ("torch objective", torch_obj), ("ops.hyper_grads", lambda: ops.hyper_grads(gobj, cw, o["dsv"], o["dmean"], None, shapes)), ("torch hyper grads", torch_hyper),
         ("softplus", lambda: torch.nn.functional.softplus(sv))]
for name, fn in cases:
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-20s host %.1f us per call   (with the device drained: %.1f us)" % (name, 1e6 * (t1 - t0) / 2000, 1e6 * (t2 - t0) / 2000), flush=True)
