"""dkt_lowrank_gram_f32 / dkt_lowrank_bwd_f32: the f16-split kernels (DKT_GRAM_UNIT_ROWS) against the exact-fp32 kernels and float64 -- errors and times.
Measurement tooling.

    python tools/lowrank_f16_ab.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.load()
p = lambda t: None if t is None else t.data_ptr()
st_ = lambda: torch.cuda.current_stream().cuda_stream


def perm(d=64):
    """storage index d' = 16 q + m  <->  column 4 m + q"""
    col = torch.arange(d)
    return (16 * (col % 4) + col // 4).to(dev)          # perm[col] = d'


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for (b, n, d, c, per_ep_y) in [(64, 105, 64, 5, False), (64, 420, 64, 20, False), (64, 91, 36, 7, True), (8192, 105, 64, 5, False), (2048, 420, 64, 20, False)]:
    g = torch.Generator(device=dev).manual_seed(n + c)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev) + 0.7, dim=2).contiguous()
    if per_ep_y:
        y = (torch.randn(b, c, n, generator=g, device=dev) * 3.0).contiguous()
        ybs = c * n
    else:
        cls = torch.arange(c, device=dev).repeat_interleave(n // c)
        y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
        ybs = 0
    mean = torch.randn(c, generator=g, device=dev) * 0.3
    pm = perm()
    zp = torch.zeros(b, n, 64, device=dev, dtype=torch.float64)
    zp[:, :, pm[:d]] = z.double()                         # permuted, zero-padded features
    r = (y.double() if per_ep_y else y.double().unsqueeze(0).expand(b, c, n)) - mean.double().view(1, c, 1)
    a_ref = zp.transpose(1, 2) @ zp
    p_ref = r @ zp                                        # [b, c, 64]
    res = {}
    for flags in (0, 1):
        a = torch.empty(b, 64, 64, device=dev)
        pp = torch.empty(b, c, 64, device=dev)
        run = lambda: _lib.check(lib.dkt_lowrank_gram_f32(p(z), p(y), ybs, p(mean), p(a), p(pp), b, c, n, d, flags, st_()), "gram")
        t = timeit(run)
        ea = ((a.double() - a_ref).abs().max() / a_ref.abs().max()).item()
        ep = ((pp.double() - p_ref).abs().max() / p_ref.abs().max()).item()
        sym = bool(torch.equal(a, a.transpose(1, 2)))
        res[flags] = (t, ea, ep, sym)
    print("gram  B=%d N=%d D=%d C=%d: fp32 %.4f ms (A err %.2e, P err %.2e, sym %s) | f16 split %.4f ms (A err %.2e, P err %.2e, sym %s)" % (
        b, n, d, c, *res[0], *res[1]), flush=True)
    # backward: dZ = g (V^T T + 2 Z W'), W' symmetric 64 x 64 (permuted order), T [b, c, 64] (permuted), V [b, c, n]
    wd = torch.randn(b, 64, 64, generator=g, device=dev) * 0.05
    wd = (wd + wd.transpose(1, 2)).contiguous()
    tt = (torch.randn(b, c, 64, generator=g, device=dev) * 0.2).contiguous()
    v = (torch.randn(b, c, n, generator=g, device=dev) * 0.01).contiguous()
    gobj = torch.linspace(0.5, 1.5, b, device=dev)
    dzp = gobj.double().view(b, 1, 1) * (v.double().transpose(1, 2) @ tt.double() + 2.0 * zp @ wd.double())      # [b, n, 64] permuted
    dz_ref = dzp[:, :, pm[:d]]
    res = {}
    for flags in (0, 1):
        dz = torch.empty(b, n, d, device=dev)
        run = lambda: _lib.check(lib.dkt_lowrank_bwd_f32(p(z), p(v), p(tt), p(wd), p(gobj), p(dz), b, c, n, d, flags, st_()), "bwd")
        t = timeit(run)
        e2 = ((dz.double() - dz_ref).norm() / dz_ref.norm()).item()
        em = ((dz.double() - dz_ref).abs().max() / dz_ref.abs().max()).item()
        res[flags] = (t, e2, em)
    print("bwd   B=%d N=%d D=%d C=%d: fp32 %.4f ms (rel-L2 %.2e, max %.2e) | f16 split %.4f ms (rel-L2 %.2e, max %.2e)" % (b, n, d, c, *res[0], *res[1]), flush=True)
