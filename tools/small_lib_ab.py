"""Same-box A/B of the N <= 32 Gram kernels between two BUILDS of the library (product vs a variant .so), e.g. one built from an older dkt_gram_small.hip with
_lib.build(out=..., replace={'dkt_gram_small.hip': path}).   python tools/small_lib_ab.py deep-kernel-transfer_amd/libdkt_prev.so"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dkt_amd = importlib.import_module("deep-kernel-transfer_amd")
ops, _lib = dkt_amd.ops, dkt_amd._lib
other = os.path.abspath(sys.argv[1])
libs = {"product": _lib.load(), os.path.basename(other): _lib.load(other)}
dev = torch.device("cuda:0")
p = lambda t: 0 if t is None else t.data_ptr()
st = lambda: torch.cuda.current_stream().cuda_stream
for (b, n, d, kind) in [(8192, 19, 2916, 1), (8192, 25, 1600, 0), (8192, 19, 512, 1), (8192, 25, 64, 0), (1024, 19, 2916, 1), (8192, 10, 2916, 1), (8192, 32, 1600, 0)]:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.randn(b, n, d, device=dev, generator=g) * 0.05
    w = torch.randn(b, n, n, device=dev, generator=g) * 0.01
    ls = torch.tensor([1.3], device=dev)
    eo = {k: torch.empty(b, n, n, device=dev) for k in libs}
    dzo = {k: torch.empty_like(z) for k in libs}
    fw, bw = {}, {}
    for rnd in range(4):
        for name, lib in libs.items():
            def f():
                assert lib.dkt_gram_f32(p(z), 0, p(eo[name]), b, n, n, d, kind, p(ls), st()) == 0
            def bk():
                assert lib.dkt_gram_bwd_f32(p(w), p(z), p(dzo[name]), b, n, d, 0, 0, st()) == 0
            for fn, acc in ((f, fw), (bk, bw)):
                for _ in range(2):
                    fn()
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(10):
                    fn()
                t1.record()
                torch.cuda.synchronize()
                acc.setdefault(name, []).append(t0.elapsed_time(t1) / 10)
    names = list(libs)
    af, ab = b * (n * d + n * n) * 4, b * (2 * n * d + n * n) * 4
    print("B=%d N=%d D=%d kind=%d  forward: " % (b, n, d, kind) + "  ".join("%s %.4f ms (%.3f)" % (k, min(v), af / min(v) / 8e9) for k, v in fw.items()) +
          ("  bitwise equal" if torch.equal(eo[names[0]], eo[names[1]]) else "  E DIFFERS") + " |  backward: " +
          "  ".join("%s %.4f ms (%.3f)" % (k, min(v), ab / min(v) / 8e9) for k, v in bw.items()) + ("  bitwise equal" if torch.equal(dzo[names[0]], dzo[names[1]]) else "  dZ DIFFERS"), flush=True)
    del z, w, eo, dzo
