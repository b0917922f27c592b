"""Same-box A/B of the level-1 Gram backward (dkt_gram_bwd_f32, unit rows) between two BUILDS: the product and a variant built with an extra macro, e.g.
    DKT_EXTRA_HIPCC_FLAGS=-DDKT_GRAM_BWD_BRANCHY_STORES python -c "import importlib; L = importlib.import_module('deep-kernel-transfer_amd')._lib; L.build(out='deep-kernel-transfer_amd/libdkt_bwd_branchy.so')"
    python tools/gram_bwd_lib_ab.py deep-kernel-transfer_amd/libdkt_bwd_branchy.so
Alternating timing rounds, outputs compared bitwise."""
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
shapes = [(8192, 105, 1600), (8192, 85, 512), (8192, 80, 1600), (8192, 128, 1600), (2048, 105, 1600), (8192, 105, 64)]
if len(sys.argv) > 4 and sys.argv[2] != "big":
    a = [int(v) for v in sys.argv[2:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
if len(sys.argv) > 2 and sys.argv[2] == "big":
    shapes = []
for (b, n, d) in shapes:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev, generator=g), dim=2)
    w = torch.randn(b, n, n, device=dev, generator=g) * 0.01
    w = (w + w.transpose(1, 2)).contiguous()
    outs = {name: torch.empty_like(z) for name in libs}
    res = {}

    def run(name):
        rc = libs[name].dkt_gram_bwd_f32(p(w), p(z), p(outs[name]), b, n, d, 0, ops.GRAM_UNIT_ROWS, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    for rnd in range(4):
        for name in libs:
            for _ in range(2):
                run(name)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                run(name)
            t1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(t0.elapsed_time(t1) / 10)
    names = list(libs)
    same = torch.equal(outs[names[0]], outs[names[1]])
    alg = b * (2 * n * d + n * n) * 4
    print("gram_bwd B=%d N=%d D=%d: " % (b, n, d) + "  ".join("%s %.4f ms (%.3f of 8 TB/s)" % (k, min(v), alg / min(v) / 1e9 / 8.0) for k, v in res.items()) +
          ("  bitwise equal" if same else "  OUTPUTS DIFFER"), flush=True)
    del z, w, outs

# large-N rows kernel (128 < N <= 448, unit rows, symmetric W): the same comparison
for (b, n, d) in [(1024, 420, 512), (1024, 320, 512), (2048, 150, 1600), (512, 256, 1600)]:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev, generator=g), dim=2)
    w = torch.randn(b, n, n, device=dev, generator=g) * 0.01
    w = (w + w.transpose(1, 2)).contiguous()
    outs = {name: torch.empty_like(z) for name in libs}
    res = {}
    for rnd in range(4):
        for name in libs:
            def run():
                rc = libs[name].dkt_gram_bwd_f32(p(w), p(z), p(outs[name]), b, n, d, 0, ops.GRAM_UNIT_ROWS | ops.GRAM_W_SYMMETRIC, torch.cuda.current_stream().cuda_stream)
                assert rc == 0
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
    same = torch.equal(outs[names[0]], outs[names[1]])
    alg = b * (2 * n * d + n * n) * 4
    print("gram_bwd large-N B=%d N=%d D=%d: " % (b, n, d) + "  ".join("%s %.4f ms (%.3f of 8 TB/s)" % (k, min(v), alg / min(v) / 1e9 / 8.0) for k, v in res.items()) +
          ("  bitwise equal" if same else "  OUTPUTS DIFFER"), flush=True)
    del z, w, outs
