"""Per-phase shader-clock breakdown of the large-N Gram backward (a -DDKT_GBW_CLOCKS build: wave 0 of every workgroup leaves its clocks in the first row
of its block of dZ).  Measurement tooling."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402

var = os.path.join(ROOT, "deep-kernel-transfer_amd", "libdkt_gbwclk.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.environ["DKT_EXTRA_HIPCC_FLAGS"] = "-DDKT_GBW_CLOCKS"
    dkt_amd._lib.build(out=var)
    sys.exit(0)
os.environ["DKT_AMD_LIB"] = var
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
names = ["prologue + first slab staged", "issue next slab's loads", "products (LDS reads + MFMA)", "dZ stores", "wait loads + split + LDS stores", "barrier"]
for (b, n, d) in [(1024, 420, 512), (1024, 320, 512)]:
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    w = torch.randn(b, n, n, generator=g, device=dev)
    w = (w + w.transpose(1, 2)).contiguous()
    for _ in range(2):
        dz = ops.gram_bwd(w, z, None, unit_rows=True, w_symmetric=True)
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    dz = ops.gram_bwd(w, z, None, unit_rows=True, w_symmetric=True)
    t.record()
    torch.cuda.synchronize()
    nrb = (n + 127) // 128
    for rb in range(nrb):
        v = dz[:, 128 * rb, :6].double().mean(0)
        tot = v.sum().item()
        both = dz[:, 128 * rb, 6:8].double().sum(0)
        print("     s_memtime / s_memrealtime (100 MHz) over the workgroups: %.1f -> shader clock %.2f GHz" % ((both[0] / both[1]).item(), (both[0] / both[1]).item() * 0.1))
        print("B=%d N=%d D=%d  kernel %.3f ms; row block %d: wave 0, mean ticks per workgroup (total %.0f = %.1f us at 2.4 GHz)" % (b, n, d, s.elapsed_time(t), rb, tot, tot / 2400.0))
        for nm, x in zip(names, v.tolist()):
            print("     %-36s %9.0f  (%4.1f %%)" % (nm, x, 100 * x / tot))
    sys.stdout.flush()
