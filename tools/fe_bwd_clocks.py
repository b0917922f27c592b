"""Shader clocks per phase of the fused front-end backward (gram_bn_bwd_ep_kernel, wave 0 of every workgroup), measurement build:
    DKT_EXTRA_HIPCC_FLAGS=-DDKT_FE_CLOCKS python -c "from dkt_amd import _lib; _lib.build(force=True, out='deep-kernel-transfer_amd/libdkt_feclk.so')"
    DKT_AMD_LIB=.../libdkt_feclk.so python tools/fe_bwd_clocks.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
c, per, d = 5, 21, 1600
n = c * per
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.randn(b, n, d, generator=g, device=dev).abs() + 1.0)
gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
e, rnorm, st = ops.gram_bn_train(x, gamma, beta, 1e-5)
w = torch.randn(b, n, n, generator=g, device=dev) * 1e-4
w = 0.5 * (w + w.transpose(1, 2)).contiguous()
for _ in range(3):
    dx, dg, db = ops.gram_bn_bwd(w, e, x, st["a"], st["s"], rnorm, st["mean"], st["rstd"], None)
torch.cuda.synchronize()
ck = dg[:, :7].double()
names = ["prologue (W / E staging, A fragments, first image)", "loads issued + MFMA loop", "wait for the epilogue operands / prefetch", "epilogue, first half",
         "split + LDS stores of the next image", "barrier wait", "epilogue, second half + dX stores"]
tot = ck.sum(1)
print("fused backward, B = %d, N = %d, D = %d: shader clocks of wave 0 per EPISODE, mean / p10 / p90 over the workgroups" % (b, n, d))
for i, nm in enumerate(names):
    v = ck[:, i]
    print("%-58s mean %9.0f  p10 %9.0f  p90 %9.0f  (%4.1f %%)" % (nm, v.mean().item(), v.quantile(0.1).item(), v.quantile(0.9).item(), 100 * v.mean().item() / tot.mean().item()))
print("%-58s mean %9.0f" % ("total", tot.mean().item()))
