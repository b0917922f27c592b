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
c = int(sys.argv[2]) if len(sys.argv) > 2 else 5
flags = 1 | (32 if len(sys.argv) > 3 and sys.argv[3] == "f32mfma" else 0)      # WANT_GRAD [| FORCE_F32MFMA: the exact-fp32 twin]
n, d = (105 // c) * c if c > 1 else 105, 64
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
e = ops.gram(z)
cls = torch.arange(c, device=dev).repeat_interleave(n // c)[:n]
y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.full((c,), 0.7, device=dev) + 0.02 * torch.arange(c, device=dev)
mean, noise = torch.zeros(c, device=dev), torch.full((c,), 0.1, device=dev)
cw = torch.full((c,), -1.0 / (c * n), device=dev)
nwg = (b + 1) // 2
ws = torch.zeros(nwg * 10 * 48, dtype=torch.int64, device=dev)
wpg = c if c <= 5 else 5
outs = dict(logp=torch.empty(b, c, device=dev), alpha=torch.empty(b, c, n, device=dev), jit=torch.empty(b, c, device=dev),
            info=torch.empty(b, c, dtype=torch.int32, device=dev), w=torch.empty(b, n, n, device=dev), dsv=torch.empty(b, c, device=dev),
            dmean=torch.empty(b, c, device=dev), dnoise=torch.empty(b, c, device=dev))
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
for it in range(3):
    st = lib.dkt_mll_f32(p(e), p(y), 0, p(sv), p(mean), p(noise), b, c, n, 1e-6, 3, flags, p(cw), p(outs["logp"]), p(outs["alpha"]), None,
                         p(outs["w"]), p(outs["dsv"]), p(outs["dmean"]), p(outs["dnoise"]), p(outs["jit"]), p(outs["info"]), p(ws),
                         ws.numel() * 8, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
torch.cuda.synchronize()
raw = ws.cpu().numpy().reshape(nwg, 10, 48)[:, :2 * wpg, :]
t = raw.astype(np.float64)
names = ["start->row0 issued", "phase 1 (factorisation)", "phase 2 (inverse)", "alpha", "phase 3 (M^T M)", "reductions + scalars",
         "LDS accumulate (turns + barriers)", "store W"]
dt = np.diff(t[:, :, :9], axis=2)
print("C = %d, N = %d, kernel %s" % (c, n, "f32mfma twin" if flags & 32 else "h2 (default)"))
print("waves %d; mean / p10 / p90 shader clocks per phase (s_memtime ticks = 100 MHz constant clock? see total)" % (nwg * 10))
for i, nm in enumerate(names):
    v = dt[:, :, i].ravel()
    print("%-36s mean %9.0f   p10 %9.0f   p90 %9.0f" % (nm, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
# phase 1 in detail, grouped by how many waves of the workgroup share the wave's SIMD (HW_ID bits 5:4)
simd = (raw[:, :, 9] >> 4) & 3
nt = (n + 1 + 15) // 16
share = np.zeros_like(simd)
for wg in range(nwg):
    cnt = np.bincount(simd[wg], minlength=4)
    share[wg] = cnt[simd[wg]]
for sh in sorted(set(share.ravel().tolist())):
    m = share == sh
    sw = [(t[:, :, 13] - t[:, :, 12])[m].mean()] + [(t[:, :, 13 + 2 * k] - t[:, :, 12 + 2 * k])[m].mean() for k in range(1, nt)]
    bu = [(t[:, :, 14 + 2 * k] - t[:, :, 13 + 2 * k])[m].mean() for k in range(nt - 1)]
    print("waves sharing their SIMD with %d of the workgroup: %d   phase 1 mean %.0f" % (sh, m.sum(), (t[:, :, 2] - t[:, :, 1])[m].mean()))
    print("   sweeps (k = 0 plain; k >= 1 interleaved with step k-1's updates): " + " ".join("%.0f" % v for v in sw))
    print("   panel + next tile row bursts:                                     " + " ".join("%.0f" % v for v in bu))
    if raw[:, :, 32].any():
        pv = [(t[:, :, 32] - t[:, :, 12])[m].mean()] + [(t[:, :, 32 + q] - t[:, :, 31 + q])[m].mean() for q in range(1, 16)]
        print("   sweep 0 by pivot (first: P1 start -> pivot 0's head done):         " + " ".join("%.0f" % v for v in pv))
tot = (t[:, :, 8] - t[:, :, 0]).ravel()
print("%-36s mean %9.0f   p10 %9.0f   p90 %9.0f" % ("wave total", tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
