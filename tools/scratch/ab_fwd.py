import os, sys, torch
sys.path.insert(0, "/root/repo")
import dkt_amd
from dkt_amd import ops
dev = torch.device("cuda", 0)
b, n, d = 8192, 105, 1600
z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev), dim=2).contiguous()
def timed(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
res = {}
for rep in range(5):
    for var in ("2223", "22232"):
        os.environ["DKT_GRAM_UNIT_VAR"] = var
        ops._sync_env(dkt_amd._lib.load())
        res.setdefault(var, []).append(timed(lambda: ops.gram(z, kind=ops.KERNEL_LINEAR_UNIT)))
for k, v in res.items(): print("fwd", k, " ".join("%.4f" % x for x in v))
os.environ.pop("DKT_GRAM_UNIT_VAR")
w = torch.randn(b, n, n, device=dev) * 0.01
res = {}
for rep in range(5):
    for var in ("222", "1222"):
        os.environ["DKT_GRAM_BWD_UNIT_VAR"] = var
        ops._sync_env(dkt_amd._lib.load())
        res.setdefault(var, []).append(timed(lambda: ops.gram_bwd(w, z, unit_rows=True)))
for k, v in res.items(): print("bwd", k, " ".join("%.4f" % x for x in v))
