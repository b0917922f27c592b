"""A/B of the cache policy of the two Gram kernels at the headline shape (8192 x 105 x 1600, unit rows): default vs non-temporal
Z loads / dZ stores, next to the arithmetic-free copy kernels of the same access pattern.  Measurement tooling."""
import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
b, n, d = 8192, 105, 1600


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev), dim=2).contiguous()
dlib = dkt_amd._lib.load_diag()
fn = dlib.dkt_diag_stream_f32
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(b, device=dev)
outz = torch.empty_like(z)
st = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    for mode, name, o, byt in ((0, "read, 128-B slab", out, 4.0), (9, "read, 128-B slab, nt loads", out, 4.0), (3, "copy, 256-B slab", outz, 8.0),
                               (5, "copy, 256-B slab, nt stores", outz, 8.0), (6, "copy, 256-B slab, nt loads", outz, 8.0),
                               (7, "copy, 256-B slab, nt both", outz, 8.0), (4, "copy, linear", outz, 8.0), (8, "copy, linear, nt both", outz, 8.0)):
        ms = timed(lambda: fn(z.data_ptr(), o.data_ptr(), b, n, d, mode, st))
        print("%-30s %.4f ms  %.0f GB/s" % (name, ms, byt * b * n * d / ms / 1e6), flush=True)
del outz
w = torch.randn(b, n, n, device=dev) * 0.01
e_ref = None
for rep in range(2):
    for var in ("2223", "22232"):
        os.environ["DKT_GRAM_UNIT_VAR"] = var
        ops._sync_env(dkt_amd._lib.load())
        e = ops.gram(z, kind=ops.KERNEL_LINEAR_UNIT)
        ms = timed(lambda: ops.gram(z, kind=ops.KERNEL_LINEAR_UNIT))
        e_ref = e if e_ref is None else e_ref
        print("gram fwd var %-6s %.4f ms  %.0f GB/s  equal %s" % (var, ms, 4.0 * b * (n * d + n * n) / ms / 1e6, bool(torch.equal(e, e_ref))), flush=True)
    os.environ.pop("DKT_GRAM_UNIT_VAR")
    g_ref = None
    for var in ("222", "1222", "2222", "3222"):
        os.environ["DKT_GRAM_BWD_UNIT_VAR"] = var
        ops._sync_env(dkt_amd._lib.load())
        g = ops.gram_bwd(w, z, unit_rows=True)
        ms = timed(lambda: ops.gram_bwd(w, z, unit_rows=True))
        g_ref = g if g_ref is None else g_ref
        print("gram bwd var %-6s %.4f ms  %.0f GB/s  equal %s" % (var, ms, 4.0 * b * (2 * n * d + n * n) / ms / 1e6, bool(torch.equal(g, g_ref))), flush=True)
    os.environ.pop("DKT_GRAM_BWD_UNIT_VAR")
ops._sync_env(dkt_amd._lib.load())
