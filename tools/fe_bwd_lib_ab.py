"""Same-box A/B of the fused backward (dkt_gram_bn_bwd_f32) between two BUILDS of the library: the product and a variant built with an extra macro, e.g.
    DKT_EXTRA_HIPCC_FLAGS=-DDKT_FE_BWD_RELOAD_X python -c "import importlib; L = importlib.import_module('deep-kernel-transfer_amd')._lib; L.build(out='deep-kernel-transfer_amd/libdkt_fe_reload.so')"
    python tools/fe_bwd_lib_ab.py deep-kernel-transfer_amd/libdkt_fe_reload.so
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
st_ = lambda: torch.cuda.current_stream().cuda_stream
for (b, n, d) in [(2048, 105, 1600), (8192, 105, 1600), (2048, 85, 512), (2048, 80, 640), (2048, 128, 1600), (4096, 105, 64)]:
    g = torch.Generator(device=dev).manual_seed(n + d + 1)
    x = torch.randn(b, n, d, device=dev, generator=g).abs() * 2.0 + 1.0
    gamma, beta = 0.5 + torch.rand(d, device=dev, generator=g), 0.2 * torch.randn(d, device=dev, generator=g)
    e, rnorm, st = ops.gram_bn_train(x, gamma, beta, 1e-5)
    w = torch.randn(b, n, n, device=dev, generator=g) * 0.01
    w = (w + w.transpose(1, 2)).contiguous()
    gobj = torch.linspace(0.5, 1.5, b, device=dev)
    outs, res = {}, {}
    for name in libs:
        outs[name] = (torch.empty_like(x), torch.empty(b, d, device=dev), torch.empty(b, d, device=dev))

    def run(name):
        dx, dg, db = outs[name]
        rc = libs[name].dkt_gram_bn_bwd_f32(p(w), p(e), p(x), p(st["a"]), p(st["s"]), d, p(st["mean"]), p(st["rstd"]), p(rnorm), p(gobj), p(dx), p(dg), p(db), b, n, d, st_())
        assert rc == 0
    for rnd in range(3):
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
    same = all(torch.equal(a, c) for a, c in zip(outs[names[0]], outs[names[1]]))
    alg = b * (2 * n * n + 2 * n * d + 4 * n + 24 * d) * 4
    print("gram_bn_bwd B=%d N=%d D=%d: " % (b, n, d) + "  ".join("%s %.4f ms (%.3f of 8 TB/s)" % (k, min(v), alg / min(v) / 1e9 / 8.0) for k, v in res.items()) +
          ("  bitwise equal" if same else "  OUTPUTS DIFFER"), flush=True)
    del x, e, w, outs
