"""Read bandwidth as a function of the working set: a linear 16-byte-per-lane read of W bytes, repeated back to back, for W from 32 MB
(inside the 256-MB memory-side cache) to 4 GB (HBM).  Answers whether re-reads that hit the memory-side cache are served faster than
the ~5-6 TB/s an HBM stream reaches (the tile-array marginal likelihood re-reads every tile 2-4 x)."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402

lib = dkt_amd._lib.load_diag()
fn = lib.dkt_diag_stream_f32
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
n, d = 16, 1024                       # 64 KB per workgroup
for mb in (32, 64, 128, 192, 256, 384, 512, 1024, 4096):
    b = mb * 1024 * 1024 // (4 * n * d)
    z = torch.randn(b, n, d, device=dev)
    out = torch.zeros(b, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    reps = max(10, 8192 // mb)
    for _ in range(3):
        fn(z.data_ptr(), out.data_ptr(), b, n, d, 1, st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn(z.data_ptr(), out.data_ptr(), b, n, d, 1, st)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print("working set %5d MB  %6d workgroups  %.4f ms per pass  %.0f GB/s" % (mb, b, ms, 4.0 * b * n * d / ms / 1e6), flush=True)
    del z
