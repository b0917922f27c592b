"""dkt_mll_f32 for N <= 127: the default f16-split wave-per-matrix kernel (dkt_mll_h2.hip) next to its exact-fp32 MFMA twin
(DKT_MLL_FORCE_F32MFMA) -- time per launch at the bench shapes and the error of both against a float64 torch reference on a few
episodes, at the initial hyper-parameters and at a large-outputscale / small-noise point.  Measurement tooling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
shapes = [(8192, 5, 105, 1600, True, 0.69, 0.1), (8192, 5, 105, 64, True, 0.69, 0.1), (8192, 5, 85, 512, True, 0.69, 0.1), (8192, 5, 25, 64, False, 0.69, 0.1),
          (8192, 5, 105, 1600, True, 30.0, 0.01), (4096, 1, 19, 2916, True, 0.69, 0.69), (1024, 20, 100, 64, True, 0.69, 0.1)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]


def ref64(e, y, sv, mean, noise, cw, nb=4):
    e = e[:nb].double(); y = y.double()
    n = e.shape[-1]
    outs = []
    for b in range(e.shape[0]):
        eb = e[b].clone().requires_grad_(True)
        logps = []
        for c in range(y.shape[0]):
            k = sv[c].double() * eb + noise[c].double() * torch.eye(n, device=dev, dtype=torch.float64)
            l = torch.linalg.cholesky(k)
            r = (y[c] - mean[c].double()).unsqueeze(1)
            a = torch.cholesky_solve(r, l)
            logps.append(-0.5 * (r * a).sum() - l.diagonal().log().sum() - 0.5 * n * 1.8378770664093453)
        lp = torch.stack(logps)
        (cw.double() * lp).sum().backward()
        outs.append((lp.detach(), eb.grad.detach()))
    return outs


for (b, c, n, d, grad, s0, nz) in shapes:
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    if cls.numel() < n:
        cls = torch.cat([cls, torch.zeros(n - cls.numel(), dtype=cls.dtype, device=dev)])
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), s0, device=dev) * (1.0 + 0.03 * torch.arange(c, device=dev))
    mean = 0.01 * torch.arange(c, device=dev, dtype=torch.float32)
    noise = torch.full((c,), nz, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    zb = min(b, 2048)
    z = torch.nn.functional.normalize(torch.randn(zb, n, d, generator=g, device=dev), dim=2).contiguous()
    e = ops.gram(z).repeat(b // zb, 1, 1).contiguous()
    res = {}
    for f32 in (False, True):
        for _ in range(3):
            out = ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=grad, force_f32mfma=f32)
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            out = ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=grad, force_f32mfma=f32)
        t.record()
        torch.cuda.synchronize()
        res[f32] = (s.elapsed_time(t) / 20, out)
    r64 = ref64(e, y, sv, mean, noise, cw)
    errs = {}
    for f32 in (False, True):
        o = res[f32][1]
        el = max(((o["logp"][i].double() - r64[i][0]).abs() / r64[i][0].abs()).max().item() for i in range(len(r64)))
        # the kernels' W is d obj / d E with E symmetric-by-halves: W + W^T / 2 conventions differ by nothing here (W symmetric = full gradient / 1)
        ew = max(((o["w"][i].double() - 0.5 * (r64[i][1] + r64[i][1].T)).norm() / r64[i][1].norm()).item() for i in range(len(r64))) if grad else 0.0
        errs[f32] = (el, ew)
    flops = c * (n ** 3 / 3 + (n ** 3 if grad else n ** 3 / 3) + 2 * n * n) * b
    print("B=%5d C=%2d N=%3d D=%4d grad=%d sv=%.2f nz=%.2g | h2 %.4f ms (%.1f TF)  f32mfma %.4f ms (%.1f TF) | logp err h2 %.2e f32 %.2e | W err h2 %.2e f32 %.2e | info %d %d" % (
        b, c, n, d, grad, s0, nz, res[False][0], flops / res[False][0] / 1e9, res[True][0], flops / res[True][0] / 1e9,
        errs[False][0], errs[True][0], errs[False][1], errs[True][1], int(res[False][1]["info"].abs().max()), int(res[True][1]["info"].abs().max())), flush=True)
