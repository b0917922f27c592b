"""Per-phase shader-clock totals of the wave-per-episode marginal-likelihood kernel (a -DDKT_MFMA_CLOCKS build of the library).
usage (GPU box): python tools/mll_h2e_clocks.py [B] [N]     Measurement tooling."""
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
os.environ["DKT_MLL_H2E_MINB"] = "1"
from dkt_amd import ops  # noqa: E402

lib = dkt_amd._lib.load()
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n = int(sys.argv[2]) if len(sys.argv) > 2 else 105
c, d = 5, 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
z = torch.nn.functional.normalize(torch.randn(min(b, 2048), n, d, generator=g, device=dev), dim=2).contiguous()
e = ops.gram(z).repeat((b + 2047) // 2048, 1, 1)[:b].contiguous()
cls = torch.arange(c, device=dev).repeat_interleave(n // c)[:n]
y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.full((c,), 0.7, device=dev) + 0.02 * torch.arange(c, device=dev)
mean, noise = torch.zeros(c, device=dev), torch.full((c,), 0.1, device=dev)
cw = torch.full((c,), -1.0 / (c * n), device=dev)
ws = torch.zeros(b * 8, dtype=torch.int64, device=dev)
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
raw = ws.cpu().numpy().reshape(b, 8)
t = raw[:, :7].astype(np.float64)
names = ["E load + kappa + form row 0 (sum over classes)", "phase 1", "phase 2", "alpha", "phase 3 (+ split, rank-one)", "scalars", "zero-init + final W store"]
print("wave-per-episode kernel, B = %d, C = %d, N = %d: shader clocks per EPISODE (5 classes), mean / p10 / p90" % (b, c, n))
for i, nm in enumerate(names):
    print("%-48s mean %9.0f   p10 %9.0f   p90 %9.0f" % (nm, t[:, i].mean(), np.percentile(t[:, i], 10), np.percentile(t[:, i], 90)))
tot = t.sum(1)
print("%-48s mean %9.0f   p10 %9.0f   p90 %9.0f   (per class matrix: %.0f)" % ("total", tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90), tot.mean() / c))
simd = (raw[:, 7] >> 4) & 3
print("waves per SIMD id:", np.bincount(simd, minlength=4))
