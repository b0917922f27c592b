"""Does running the HBM-bound Gram kernels of one chunk of episodes concurrently with the VALU-bound MLL kernel of another
chunk (separate HIP streams) shorten the training step?  Measurement tooling; prints only."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B, N, D, C = 8192, 105, 1600, 5
z = torch.nn.functional.normalize(torch.randn(B, N, D, device=dev), dim=2).contiguous()
cls = torch.arange(C, device=dev).repeat_interleave(N // C)
y = torch.where(cls.unsqueeze(0) == torch.arange(C, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.full((C,), 0.69, device=dev) + 0.01 * torch.arange(C, device=dev)
mean = torch.zeros(C, device=dev)
noise = torch.full((C,), 0.1, device=dev)
cw = torch.full((C,), -1.0 / (C * N), device=dev)
g = torch.full((B,), 1.0 / B, device=dev)


def chain(zk, gk):
    e = ops.gram(zk, None, ops.KERNEL_LINEAR_UNIT)
    out = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
    dz = ops.gram_bwd(out["w"], zk, gk, unit_rows=True)
    return out["logp"], dz


def run(nchunk, nstream):
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    cur = torch.cuda.current_stream()
    zs, gs = z.chunk(nchunk), g.chunk(nchunk)

    def step():
        outs = []
        if nstream == 1:
            for k in range(nchunk):
                outs.append(chain(zs[k], gs[k]))
            return outs
        ev = torch.cuda.Event()
        ev.record(cur)
        for s in streams:
            s.wait_event(ev)
        for k in range(nchunk):
            with torch.cuda.stream(streams[k % nstream]):
                outs.append(chain(zs[k], gs[k]))
        for s in streams:
            e2 = torch.cuda.Event()
            e2.record(s)
            cur.wait_event(e2)
        return outs

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        outs = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    return dt, outs


ref = None
for nchunk, nstream in [(1, 1), (4, 1), (2, 2), (4, 2), (8, 2), (4, 4), (8, 4), (16, 2), (16, 4), (3, 3), (6, 3)]:
    dt, outs = run(nchunk, nstream)
    lp = torch.cat([o[0] for o in outs])
    dz = torch.cat([o[1] for o in outs])
    if ref is None:
        ref = (lp, dz)
    same = bool(torch.equal(lp, ref[0]) and torch.equal(dz, ref[1]))
    print("chunks %2d streams %d: %.3f ms/step  %.0f episodes/s  bitwise equal to the single launch: %s" % (nchunk, nstream, 1e3 * dt, B / dt, same), flush=True)
