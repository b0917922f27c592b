"""Large-N symmetric Gram (unit rows): the episode-resident kernel against the 64 x 64-tile kernel of round 2 at the cfg4 shapes.  Measurement tooling."""
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
for (b, n, d) in [(1024, 420, 512), (1024, 320, 512), (1024, 190, 512), (256, 420, 1600)]:
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    for name, env in 2 * (("episode-resident (default)", {}), ("64 x 64 tiles (round 2)", {"DKT_GRAM_BIG_EP": "0"})):
        os.environ.update(env)
        try:
            for _ in range(10):
                ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
            torch.cuda.synchronize()
            s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
            t.record()
            torch.cuda.synchronize()
        finally:
            for k in env:
                os.environ.pop(k)
        ms = s.elapsed_time(t) / 10
        gb = 4.0 * (n * d + n * n) * b / 1e9
        print("B=%d N=%d D=%d  %-28s %.3f ms  (%.2f of 8 TB/s on %.2f GB)" % (b, n, d, name, ms, gb / ms / 8.0, gb), flush=True)
