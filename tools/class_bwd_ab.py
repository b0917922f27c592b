"""Same-box A/B of the class-kernel chain rule at N <= 128 (dkt_class_kernel_bwd_f32; twins library): DKT_CLASS_BWD_N128 = 0 (the element-by-element
kernel of round 3) / 1 (all loads of a row in flight, constants and partials in registers; round 5).   python tools/class_bwd_ab.py"""
import importlib
import os
import sys

os.environ["DKT_TWINS"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("deep-kernel-transfer_amd").ops
dev = torch.device("cuda:0")
for (b, c, n, cmap, power) in [(2048, 5, 105, ops.CLASSMAP_RBF, 1), (2048, 5, 105, ops.CLASSMAP_MATERN25, 1), (2048, 5, 105, ops.CLASSMAP_POLY, 2), (2048, 5, 80, ops.CLASSMAP_RBF, 1),
                              (8192, 5, 25, ops.CLASSMAP_RBF, 1), (64, 5, 105, ops.CLASSMAP_RBF, 1), (1, 5, 105, ops.CLASSMAP_RBF, 1), (1024, 8, 128, ops.CLASSMAP_RBF, 1)]:
    g = torch.Generator(device=dev).manual_seed(n + c)
    base = torch.rand(b, n, n, device=dev, generator=g) * 2.0
    base = 0.5 * (base + base.transpose(1, 2)).contiguous()
    w = torch.randn(b, c, n, n, device=dev, generator=g) * 0.01
    w = (w + w.transpose(2, 3)).contiguous()
    param = torch.linspace(0.8, 1.6, c, device=dev)
    res, outs = {}, {}
    for rnd in range(3):
        for v in ("0", "1"):
            os.environ["DKT_CLASS_BWD_N128"] = v
            for _ in range(2):
                o = ops.class_kernel_bwd(w, base, cmap, power, param)
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(10):
                o = ops.class_kernel_bwd(w, base, cmap, power, param)
            t1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(t0.elapsed_time(t1) / 10)
            outs[v] = o
    del os.environ["DKT_CLASS_BWD_N128"]
    d1 = ((outs["0"][0] - outs["1"][0]).abs().max() / outs["0"][0].abs().max()).item()
    d2 = ((outs["0"][1] - outs["1"][1]).abs().max() / outs["0"][1].abs().max()).item()
    alg = b * (c + 2) * n * n * 4
    print("B=%d C=%d N=%d map=%d: round-3 kernel %.4f ms (%.3f of 8 TB/s)  new %.4f ms (%.3f)  Wp rel diff %.1e  dparam rel diff %.1e"
          % (b, c, n, cmap, min(res["0"]), alg / min(res["0"]) / 8e9, min(res["1"]), alg / min(res["1"]) / 8e9, d1, d2), flush=True)
