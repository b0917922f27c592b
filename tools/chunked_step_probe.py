"""Does the memory-side cache (MALL, 256 MB) serve the Gram backward's second read of Z when the step runs chunk by chunk
(Gram -> MLL -> Gram backward per chunk of episodes) instead of kernel by kernel over the whole batch?  Measurement tooling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
b, n, d, c = 8192, 105, 1600, 5
g = torch.Generator(device=dev).manual_seed(0)
z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
cls = torch.arange(c, device=dev).repeat_interleave(n // c)
y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.full((c,), 0.7, device=dev); mean = torch.zeros(c, device=dev); noise = torch.full((c,), 0.1, device=dev)
cw = torch.full((c,), -1.0 / (c * n), device=dev)
scale = torch.full((b,), 1.0 / b, device=dev)


def step(chunk, order):
    outs = []
    if order == "chunked":
        for s in range(0, b, chunk):
            zc = z[s:s + chunk]
            e = ops.gram(zc, kind=ops.KERNEL_LINEAR_UNIT)
            o = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
            outs.append(ops.gram_bwd(o["w"], zc, scale[s:s + chunk], unit_rows=True, w_symmetric=True))
    else:                      # the three kernels each over the whole batch, launched chunk by chunk (same launch count, no locality)
        es = [ops.gram(z[s:s + chunk], kind=ops.KERNEL_LINEAR_UNIT) for s in range(0, b, chunk)]
        os_ = [ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw) for e in es]
        outs = [ops.gram_bwd(o["w"], z[s:s + chunk], scale[s:s + chunk], unit_rows=True, w_symmetric=True) for o, s in zip(os_, range(0, b, chunk))]
    return outs


for chunk in (8192, 2048, 1024, 512, 256, 128):
    for order in ("chunked", "kernelwise"):
        for _ in range(2):
            step(chunk, order)
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(5):
            step(chunk, order)
        s1.record()
        torch.cuda.synchronize()
        print("chunk %5d  %-10s  %.3f ms per 8192 episodes" % (chunk, order, s0.elapsed_time(s1) / 5), flush=True)
