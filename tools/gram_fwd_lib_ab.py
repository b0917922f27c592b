"""Same-box A/B of the level-1 Gram forward (dkt_gram_f32, unit rows) between two BUILDS of the library.   python tools/gram_fwd_lib_ab.py other.so [B N D]..."""
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
shapes = [(8192, 128, 1600), (8192, 120, 1600), (8192, 113, 512), (2048, 128, 1600), (8192, 128, 64), (8192, 105, 1600)]
if len(sys.argv) > 4:
    a = [int(v) for v in sys.argv[2:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
for (b, n, d) in shapes:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev, generator=g), dim=2)
    outs = {name: torch.empty(b, n, n, device=dev) for name in libs}
    res = {}
    for rnd in range(4):
        for name, lib in libs.items():
            def run():
                assert lib.dkt_gram_f32(p(z), 0, p(outs[name]), b, n, n, d, ops.KERNEL_LINEAR_UNIT, 0, torch.cuda.current_stream().cuda_stream) == 0
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
    alg = b * (n * d + n * n) * 4
    print("gram fwd B=%d N=%d D=%d: " % (b, n, d) + "  ".join("%s %.4f ms (%.3f of 8 TB/s)" % (k, min(v), alg / min(v) / 1e9 / 8.0) for k, v in res.items()) +
          ("  bitwise equal" if torch.equal(outs[names[0]], outs[names[1]]) else "  OUTPUTS DIFFER"), flush=True)
    del z, outs
