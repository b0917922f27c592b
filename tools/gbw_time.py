"""Large-N Gram backward: the default kernel (128-row blocks) against the 64-row kernel of round 2 at the cfg4 shapes.
Measurement tooling."""
import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
for (b, n, d) in [(1024, 420, 512), (1024, 320, 512), (1024, 256, 512), (1024, 190, 512), (1024, 150, 512), (2048, 190, 64)]:
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    w = torch.randn(b, n, n, generator=g, device=dev)
    w = (w + w.transpose(1, 2)).contiguous()
    for name, env in 2 * (("128-row blocks (default)", {}), ("64-row blocks (round 2)", {"DKT_GRAM_BWD_ROWS8": "0"})):
        os.environ.update(env)
        try:
            for _ in range(10):
                ops.gram_bwd(w, z, None, unit_rows=True, w_symmetric=True)
            torch.cuda.synchronize()
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.gram_bwd(w, z, None, unit_rows=True, w_symmetric=True)
            t.record()
            torch.cuda.synchronize()
        finally:
            for k in env:
                os.environ.pop(k)
        ms = s.elapsed_time(t) / 10
        gb = 4.0 * (2 * n * d + n * n) * b / 1e9
        print("B=%d N=%d D=%d  %-28s %.3f ms  (%.2f of 8 TB/s on %.2f GB)" % (b, n, d, name, ms, gb / ms / 8.0, gb), flush=True)
