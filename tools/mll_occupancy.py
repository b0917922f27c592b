"""Time dkt_mll_f32 (grad) at episode counts that put 1..4 workgroups on each CU: separates per-workgroup latency
from throughput.  Measurement tooling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

c, n, d = 5, 105, 1600
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
cls = torch.arange(c, device=dev).repeat_interleave(n // c)
y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.full((c,), 0.7, device=dev) + 0.1 * torch.arange(c, device=dev)
mean = torch.zeros(c, device=dev)
noise = torch.full((c,), 0.1, device=dev)
cw = torch.full((c,), -1.0 / (c * n), device=dev)
for b in (64, 256, 512, 768, 1024, 1280, 2048, 4096):
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    e = ops.gram(z)
    for grad in (True, False):
        for _ in range(3):
            ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=grad)
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=grad)
        t.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(t) / 20
        print("B=%5d grad=%d  %.4f ms   %.3f us/episode   (%.2f WG/CU)" % (b, grad, ms, 1e3 * ms / b, b / 256.0), flush=True)
