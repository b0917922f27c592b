"""HBM read ceiling of the Gram kernels' access pattern (arithmetic removed) vs a linear read of the same bytes."""
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
for (b, n, d) in [(2048, 105, 1600), (4096, 105, 1600), (8192, 105, 1600)]:
    z = torch.randn(b, n, d, device=dev)
    out = torch.zeros(b, device=dev)
    outz = torch.empty_like(z)
    st = torch.cuda.current_stream().cuda_stream
    for mode, name in ((0, "slab 128 B"), (2, "slab 256 B"), (1, "linear")):
        for _ in range(3):
            fn(z.data_ptr(), out.data_ptr(), b, n, d, mode, st)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn(z.data_ptr(), out.data_ptr(), b, n, d, mode, st)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print("B=%d %-11s %.4f ms  %.0f GB/s" % (b, name, ms, 4.0 * b * n * d / ms / 1e6), flush=True)
    # read + write (the Gram-backward traffic: Z read once, dZ written once)
    for mode, name in ((3, "copy, 256-B slab pattern"), (4, "copy, linear")):
        for _ in range(3):
            fn(z.data_ptr(), outz.data_ptr(), b, n, d, mode, st)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn(z.data_ptr(), outz.data_ptr(), b, n, d, mode, st)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        print("B=%d %-26s %.4f ms  %.0f GB/s (read + write)" % (b, name, ms, 8.0 * b * n * d / ms / 1e6), flush=True)
