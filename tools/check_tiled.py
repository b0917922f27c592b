"""Tile-array marginal-likelihood path (N > 127, csrc/dkt_mll_tiled.hip) against float64 torch and against the blocked twin
(DKT_MLL_FORCE_BLOCKED), plus timings.  Measurement / bring-up tooling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)


def ref64(e, y, sv, mean, noise, cw):
    # float64 on the CPU (the batched float64 Cholesky of this torch build returned garbage on the GPU from N = 320 on)
    b, n, _ = e.shape
    c = sv.numel()
    e = e.double().cpu().requires_grad_(True)
    sv64, m64, nz64 = (x.double().cpu().requires_grad_(True) for x in (sv, mean, noise))
    y, cw = y.cpu(), cw.cpu()
    k = sv64.view(1, c, 1, 1) * e.unsqueeze(1) + nz64.view(1, c, 1, 1) * torch.eye(n, dtype=torch.float64)
    r = (y.double().unsqueeze(0) - m64.view(1, c, 1)).expand(b, c, n)
    L = torch.linalg.cholesky(k)
    al = torch.cholesky_solve(r.unsqueeze(-1), L).squeeze(-1)
    logp = -0.5 * (r * al).sum(-1) - torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1) - 0.5 * n * 1.8378770664093453
    obj = (logp * cw.double().view(1, c)).sum()
    obj.backward()
    return tuple(x.to(dev) for x in (logp.detach(), al.detach(), e.grad, sv64.grad, m64.grad, nz64.grad))


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    t.record()
    torch.cuda.synchronize()
    return s.elapsed_time(t) / reps


shapes = [(6, 3, 130, 48), (5, 5, 190, 64), (4, 20, 320, 128), (3, 20, 420, 128), (2, 1, 446, 64), (7, 2, 257, 64)]
for (b, c, n, d) in shapes:
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    if cls.numel() < n:
        cls = torch.cat([cls, torch.zeros(n - cls.numel(), dtype=cls.dtype, device=dev)])
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.7, device=dev) + 0.05 * torch.arange(c, device=dev)
    mean = 0.01 * torch.arange(c, device=dev, dtype=torch.float32)
    noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    e = ops.gram(z)
    lp, al, ge, gsv, gm, gnz = ref64(e, y, sv, mean, noise, cw)
    for name, kw in (("tiled", {}), ("blocked", dict(force_blocked=True))):
        o = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, **kw)
        of = ops.mll(e, y, sv, mean, noise, want_grad=False, cls_weight=cw, **kw)
        rel = lambda x, r: ((x.double() - r).norm() / r.norm()).item()
        dsv = (o["dsv"].double() * cw.double().view(1, c)).sum(0)
        dm = (o["dmean"].double() * cw.double().view(1, c)).sum(0)
        dnz = (o["dnoise"].double() * cw.double().view(1, c)).sum(0)
        print("%-8s B=%d C=%2d N=%3d  logp %.2e (fwd-only %.2e)  alpha %.2e  W %.2e  dsv %.2e dmean %.2e dnoise %.2e  info %d  Wsym %s" % (
            name, b, c, n, ((o["logp"].double() - lp).abs() / lp.abs()).max().item(), ((of["logp"].double() - lp).abs() / lp.abs()).max().item(),
            rel(o["alpha"], al), rel(o["w"], ge), rel(dsv, gsv), rel(dm, gm), rel(dnz, gnz), int(o["info"].abs().max()),
            bool(torch.equal(o["w"], o["w"].transpose(1, 2)))), flush=True)

# a failing matrix (rank-deficient E, zero noise): info must be set, W poisoned, and the fix-up pass must agree with the blocked twin
b, c, n = 3, 2, 200
z = torch.randn(b, n, 16, generator=g, device=dev)
z[1, 150:] = z[1, :50]
e = ops.gram(z)
y = torch.where(torch.arange(n, device=dev).unsqueeze(0) % c == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.ones(c, device=dev); mean = torch.zeros(c, device=dev); noise = torch.full((c,), 1e-4, device=dev)
cw = torch.full((c,), -1.0 / (c * n), device=dev)
ot = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
ob = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, force_blocked=True)
print("failure case: info tiled %s blocked %s  jitter tiled %s blocked %s  logp diff %.2e" % (
    ot["info"].flatten().tolist(), ob["info"].flatten().tolist(), ot["jitter"].flatten().tolist(), ob["jitter"].flatten().tolist(),
    (ot["logp"] - ob["logp"]).abs().nan_to_num(0).max().item()), flush=True)

if len(sys.argv) > 1:
    for (b, c, n, d) in [(256, 20, 320, 128), (256, 20, 420, 128), (1024, 20, 420, 128), (1024, 20, 320, 128), (1024, 5, 190, 64)]:
        per = n // c
        cls = torch.arange(c, device=dev).repeat_interleave(per)
        y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
        sv = torch.full((c,), 0.7, device=dev); mean = torch.zeros(c, device=dev); noise = torch.full((c,), 0.1, device=dev)
        cw = torch.full((c,), -1.0 / (c * n), device=dev)
        z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
        e = ops.gram(z)
        flops = c * (n ** 3 / 3 + n ** 3 + 2 * n * n) * b
        tt = timed(lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw))
        tb = timed(lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, force_blocked=True))
        tf = timed(lambda: ops.mll(e, y, sv, mean, noise, want_grad=False, cls_weight=cw))
        print("B=%4d C=%2d N=%3d   tiled %.3f ms (%.1f TF)   blocked %.3f ms (%.1f TF)   tiled forward-only %.3f ms" % (
            b, c, n, tt, flops / tt / 1e9, tb, flops / tb / 1e9, tf), flush=True)
