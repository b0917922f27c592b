"""Band path of the marginal likelihood (shared base matrix, C <= 32, 128 <= N <= 432; the default from 12 classes and 192 episodes per call; csrc/dkt_mll_band.hip; named here through force_band) against float64 torch and against its
tile-array twin (force_tiled), plus timings.  Measurement / bring-up tooling.

    python tools/check_band.py            parity on a few shapes
    python tools/check_band.py time       + timings at the cfg4 shapes
"""
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


def problem(b, c, n, d, corr=False):
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    if cls.numel() < n:
        cls = torch.cat([cls, torch.zeros(n - cls.numel(), dtype=cls.dtype, device=dev)])
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.5, device=dev) + 0.04 * torch.arange(c, device=dev)
    mean = 0.01 * torch.arange(c, device=dev, dtype=torch.float32)
    noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    zr = torch.randn(b, n, d, generator=g, device=dev)
    if corr:
        cm = torch.randn(b, c, d, generator=g, device=dev)
        zr = 0.9 * cm[:, cls] + 0.1 * zr
    z = torch.nn.functional.normalize(zr, dim=2).contiguous()
    return ops.gram(z), y, sv, mean, noise, cw


shapes = [(3, 8, 128, 48, False), (2, 20, 320, 128, False), (2, 20, 420, 512, False), (2, 10, 150, 64, True), (2, 32, 432, 96, False), (3, 20, 420, 512, True), (2, 16, 257, 64, False)]
for (b, c, n, d, corr) in shapes:
    e, y, sv, mean, noise, cw = problem(b, c, n, d, corr)
    lp, al, ge, gsv, gm, gnz = ref64(e, y, sv, mean, noise, cw)
    for name, kw in (("band", dict(force_band=True)), ("tiled", dict(force_tiled=True))):
        o = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, **kw)
        of = ops.mll(e, y, sv, mean, noise, want_grad=False, cls_weight=cw, **kw)
        rel = lambda x, r: ((x.double() - r).norm() / r.norm()).item()        # noqa: E731
        dsv = (o["dsv"].double() * cw.double().view(1, c)).sum(0)
        dm = (o["dmean"].double() * cw.double().view(1, c)).sum(0)
        dnz = (o["dnoise"].double() * cw.double().view(1, c)).sum(0)
        print("%-6s B=%d C=%2d N=%3d corr=%d  logp %.2e (fwd-only %.2e, alpha fwd-only %.2e)  alpha %.2e  W %.2e  dsv %.2e dmean %.2e dnoise %.2e  info %d  Wsym %s" % (
            name, b, c, n, corr, ((o["logp"].double() - lp).abs() / lp.abs()).max().item(), ((of["logp"].double() - lp).abs() / lp.abs()).max().item(),
            rel(of["alpha"], al), rel(o["alpha"], al), rel(o["w"], ge), rel(dsv, gsv), rel(dm, gm), rel(dnz, gnz), int(o["info"].abs().max()),
            bool(torch.equal(o["w"], o["w"].transpose(1, 2)))), flush=True)

# a failing class (rank-deficient E, zero noise in one class): info / jitter must match the tile-array twin (the generic kernel redoes the episode)
b, c, n = 3, 8, 160
z = torch.randn(b, n, 16, generator=g, device=dev)
e = ops.gram(z)
y = torch.where(torch.arange(n, device=dev).unsqueeze(0) % c == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.ones(c, device=dev); mean = torch.zeros(c, device=dev); noise = torch.full((c,), 1e-2, device=dev)
noise[3] = 0.0
cw = torch.full((c,), -1.0 / (c * n), device=dev)
ot = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, force_band=True)
ob = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, force_tiled=True)
print("failure case: info band %s tiled %s\n  jitter band %s tiled %s  logp diff %.2e  W diff %.2e" % (
    ot["info"].flatten().tolist(), ob["info"].flatten().tolist(), ot["jitter"].flatten().tolist(), ob["jitter"].flatten().tolist(),
    ((ot["logp"] - ob["logp"]).abs() / ob["logp"].abs()).nan_to_num(0).max().item(), ((ot["w"] - ob["w"]).norm() / ob["w"].norm()).item()), flush=True)

if len(sys.argv) > 1:
    for (b, c, n, d) in [(1024, 20, 420, 512), (1024, 20, 320, 512), (256, 20, 420, 512), (64, 20, 420, 512), (1, 20, 420, 512), (1024, 10, 200, 64), (1024, 8, 128, 64), (1024, 32, 432, 64)]:
        e, y, sv, mean, noise, cw = problem(b, c, n, d)
        tt = timed(lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, force_band=True))
        tb = timed(lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw, force_tiled=True))
        tf = timed(lambda: ops.mll(e, y, sv, mean, noise, want_grad=False, cls_weight=cw, force_band=True))
        print("B=%4d C=%2d N=%3d   band %.3f ms   tiled %.3f ms   band forward-only %.3f ms" % (b, c, n, tt, tb, tf), flush=True)
