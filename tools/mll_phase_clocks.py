"""Per-phase shader-clock breakdown of the MFMA marginal-likelihood kernel (a -DDKT_MFMA_CLOCKS build of the library).
usage (GPU box): python tools/mll_phase_clocks.py [B]     Measurement tooling."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402

var = os.path.join(ROOT, "deep-kernel-transfer_amd", "libdkt_clk.so")      # prebuilt in the build container, travels with gpurun
os.environ["DKT_EXTRA_HIPCC_FLAGS"] = "-DDKT_MFMA_CLOCKS"
dkt_amd._lib.build(out=var)
os.environ["DKT_AMD_LIB"] = var
from dkt_amd import ops  # noqa: E402

lib = dkt_amd._lib.load()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
c, n, d = 5, 105, 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
e = ops.gram(z)
cls = torch.arange(c, device=dev).repeat_interleave(n // c)
y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.full((c,), 0.7, device=dev) + 0.02 * torch.arange(c, device=dev)
mean, noise = torch.zeros(c, device=dev), torch.full((c,), 0.1, device=dev)
cw = torch.full((c,), -1.0 / (c * n), device=dev)
nwg = (b + 1) // 2
ws = torch.zeros(nwg * 10 * 12, dtype=torch.int64, device=dev)
outs = dict(logp=torch.empty(b, c, device=dev), alpha=torch.empty(b, c, n, device=dev), jit=torch.empty(b, c, device=dev),
            info=torch.empty(b, c, dtype=torch.int32, device=dev), w=torch.empty(b, n, n, device=dev), dsv=torch.empty(b, c, device=dev),
            dmean=torch.empty(b, c, device=dev), dnoise=torch.empty(b, c, device=dev))
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
for it in range(3):
    st = lib.dkt_mll_f32(p(e), p(y), 0, p(sv), p(mean), p(noise), b, c, n, 1e-6, 3, 1, p(cw), p(outs["logp"]), p(outs["alpha"]), None,
                         p(outs["w"]), p(outs["dsv"]), p(outs["dmean"]), p(outs["dnoise"]), p(outs["jit"]), p(outs["info"]), p(ws),
                         ws.numel() * 8, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
torch.cuda.synchronize()
t = ws.cpu().numpy().reshape(nwg, 10, 12).astype(np.float64)
names = ["start->row0 issued", "phase 1 (factorisation)", "phase 2 (inverse)", "alpha", "phase 3 (M^T M)", "reductions + scalars",
         "LDS accumulate (turns + barriers)", "store W"]
dt = np.diff(t[:, :, :9], axis=2)
print("waves %d; mean / p10 / p90 shader clocks per phase (s_memtime ticks = 100 MHz constant clock? see total)" % (nwg * 10))
for i, nm in enumerate(names):
    v = dt[:, :, i].ravel()
    print("%-36s mean %9.0f   p10 %9.0f   p90 %9.0f" % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
for i, nm in ((9, "  of phase 1: all 7 sweeps"), (10, "  first sweep (waits for E)"), (11, "  last sweep")):
    v = t[:, :, i].ravel()
    print("%-36s mean %9.0f   p10 %9.0f   p90 %9.0f" % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
tot = (t[:, :, 8] - t[:, :, 0]).ravel()
print("%-36s mean %9.0f   p10 %9.0f   p90 %9.0f" % ("wave total", tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
