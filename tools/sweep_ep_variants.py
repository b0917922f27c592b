"""Sweep of the episode-resident Gram / Gram-backward variants (DKT_GRAM_UNIT_VAR / DKT_GRAM_BWD_UNIT_VAR) at the small-D shapes (cfg1: D = 64, cfg3: D = 512): the
defaults were tuned at the headline shape (D = 1600).  Measurement tooling."""
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


def t(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / 20


for (b, n, d) in [(8192, 105, 64), (8192, 85, 512), (8192, 105, 1600)]:
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    w = torch.randn(b, n, n, generator=g, device=dev)
    w = (w + w.transpose(1, 2)).contiguous()
    res = []
    for v in (2223, 22232, 2213, 2115, 211, 212, 2611, 26113, 26114, 26122):
        os.environ["DKT_GRAM_UNIT_VAR"] = str(v)
        res.append((t(lambda: ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)), v))
    os.environ.pop("DKT_GRAM_UNIT_VAR")
    print("N=%d D=%d gram:     " % (n, d) + "  ".join("%d: %.3f" % (v, ms) for ms, v in res), flush=True)
    res = []
    for v in (1222, 222, 2222, 3222, 221, 212, 211):
        os.environ["DKT_GRAM_BWD_UNIT_VAR"] = str(v)
        res.append((t(lambda: ops.gram_bwd(w, z, None, unit_rows=True, w_symmetric=True)), v))
    os.environ.pop("DKT_GRAM_BWD_UNIT_VAR")
    print("N=%d D=%d gram_bwd: " % (n, d) + "  ".join("%d: %.3f" % (v, ms) for ms, v in res), flush=True)
