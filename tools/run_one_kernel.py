"""Launch ONE hot-path kernel a few times (for rocprofv3 PMC passes).  usage: run_one_kernel.py mll|gram|gram_bwd [B [C N D]]   (default: the cfg2 shape;
1024 20 420 512 = cfg4: the Gram kernels then run with the unit-row / symmetric-W promises the training step gives them)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "mll"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
c, n, d = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (5, 105, 1600)
big = n > 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
e = ops.gram(z)
cls = torch.arange(c, device=dev).repeat_interleave(n // c)
y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.full((c,), 0.7, device=dev) + 0.1 * torch.arange(c, device=dev)
mean = torch.zeros(c, device=dev)
noise = torch.full((c,), 0.1, device=dev)
cw = torch.full((c,), -1.0 / (c * n), device=dev)
w = torch.randn(b, n, n, generator=g, device=dev)
if big:
    w = (w + w.transpose(1, 2)).contiguous()
for _ in range(5):
    if which == "mll":
        out = ops.mll(e, y, sv, mean, noise, cls_weight=cw, want_grad=True)
    elif which == "gram":
        out = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT) if big else ops.gram(z)
    else:
        out = ops.gram_bwd(w, z, None, unit_rows=big, w_symmetric=big)
torch.cuda.synchronize()
print("done", which)
