"""A/B of the pipeline variants of the unit-row (f16-split) episode-resident Gram kernel at the headline shape (DKT_GRAM_UNIT_VAR,
DESIGN.md appendix): alternating runs, 30 launches each.  Measurement tooling."""
import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dkt_amd
from dkt_amd import ops
dev = torch.device("cuda", 0)
b, n, d = 8192, 105, 1600
z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev), dim=2).contiguous()
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
res = {}
for rep in range(3):
    for var in ("2223", "2213", "26113", "26114", "26122", "212", "2115", "2611"):
        os.environ["DKT_GRAM_UNIT_VAR"] = var
        ops._sync_env(dkt_amd._lib.load())
        res.setdefault(var, []).append(timed(lambda: ops.gram(z, kind=ops.KERNEL_LINEAR_UNIT)))
for k, v in res.items(): print("fwd", k, " ".join("%.4f" % x for x in v))
