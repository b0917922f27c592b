"""Time the unit-row Gram forward / backward of the cfg2 shape with the library DKT_AMD_LIB names (same-box A/B of two builds).  python tools/experiments/lib_ab_gram.py"""
import os
import sys

os.environ["DKT_TWINS"] = "0"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
b, n, d = 8192, 105, 1600
z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
w = torch.randn(b, n, n, generator=g, device=dev)
w = (w + w.transpose(1, 2)).contiguous()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


f = min(timed(lambda: ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)) for _ in range(3))
bw = min(timed(lambda: ops.gram_bwd(w, z, None, unit_rows=True, w_symmetric=True)) for _ in range(3))
print("%s: forward %.4f ms  backward %.4f ms" % (os.path.basename(os.environ.get("DKT_AMD_LIB", "libdkt_hip.so")), f, bw), flush=True)
