"""Same-box A/B of dkt_class_kernel_bwd_f32 at N > 128 between two BUILDS of the library (product vs a variant .so, e.g. built with
DKT_EXTRA_HIPCC_FLAGS=-DDKT_CLASS_V4_GROUPWISE_LOADS).   python tools/class_bwd_lib_ab.py other.so"""
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
for (b, c, n, cmap) in [(64, 20, 420, 0), (64, 20, 320, 0), (256, 10, 200, 0), (256, 5, 150, 0), (64, 20, 420, 1), (64, 20, 420, 2), (8, 20, 420, 0)]:
    g = torch.Generator(device=dev).manual_seed(n + c)
    base = torch.rand(b, n, n, device=dev, generator=g) * 2.0
    base = 0.5 * (base + base.transpose(1, 2)).contiguous()
    w = torch.randn(b, c, n, n, device=dev, generator=g) * 0.01
    w = (w + w.transpose(2, 3)).contiguous()
    param = torch.linspace(0.8, 1.6, c, device=dev)
    res, outs = {}, {}
    for name, lib in libs.items():
        ns = int(lib.dkt_class_kernel_bwd_nsplit(b, n))
        outs[name] = (torch.empty_like(base), torch.empty(b, ns, c, device=dev))
    for rnd in range(4):
        for name, lib in libs.items():
            def run():
                assert lib.dkt_class_kernel_bwd_f32(p(w), p(base), cmap, p(param), 2, p(outs[name][0]), p(outs[name][1]), b, c, n, torch.cuda.current_stream().cuda_stream) == 0
            for _ in range(2):
                run()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                run()
            t1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(t0.elapsed_time(t1) / 10)
    names = list(libs)
    same = all(torch.equal(x, y) for x, y in zip(outs[names[0]], outs[names[1]]))
    alg = b * (c + 2) * n * n * 4
    print("class_kernel_bwd B=%d C=%d N=%d map=%d: " % (b, c, n, cmap) + "  ".join("%s %.4f ms (%.3f of 8 TB/s)" % (k, min(v), alg / min(v) / 8e9) for k, v in res.items()) +
          ("  bitwise equal" if same else "  OUTPUTS DIFFER"), flush=True)
