"""Same-box A/B of the N <= 32 Gram kernels (dkt_gram_small.hip; twins library): DKT_GRAM_SMALL_WG = 0 (a wave per task) / 1 (a workgroup per task, the four
waves interleave 64-feature blocks of the same rows).  cfg0 = the QMUL regression head: 19 rows of 2916 features.   python tools/small_wg_ab.py"""
import importlib
import os
import sys

os.environ["DKT_TWINS"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("deep-kernel-transfer_amd").ops
dev = torch.device("cuda:0")


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        out = fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps, out


for (b, n, d, kind) in [(8192, 19, 2916, ops.KERNEL_RBF), (8192, 19, 2916, ops.KERNEL_LINEAR), (8192, 25, 1600, ops.KERNEL_LINEAR), (8192, 19, 512, ops.KERNEL_RBF),
                        (8192, 25, 64, ops.KERNEL_LINEAR), (1024, 19, 2916, ops.KERNEL_RBF), (8192, 10, 2916, ops.KERNEL_RBF)]:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.randn(b, n, d, device=dev, generator=g) * 0.05
    w = torch.randn(b, n, n, device=dev, generator=g) * 0.01
    ls = torch.tensor([1.3], device=dev)
    fw, bw, outs = {}, {}, {}
    for rnd in range(3):
        for v in ("0", "1"):
            os.environ["DKT_GRAM_SMALL_WG"] = v
            ms, e = timed(lambda: ops.gram(z, None, kind, ls if kind != ops.KERNEL_LINEAR else None))
            fw.setdefault(v, []).append(ms)
            ms, dz = timed(lambda: ops.gram_bwd(w, z))
            bw.setdefault(v, []).append(ms)
            outs[v] = (e, dz)
    del os.environ["DKT_GRAM_SMALL_WG"]
    de = ((outs["0"][0] - outs["1"][0]).abs().max() / outs["0"][0].abs().max()).item()
    same_bwd = torch.equal(outs["0"][1], outs["1"][1])
    af, ab = b * (n * d + n * n) * 4, b * (2 * n * d + n * n) * 4
    print("B=%d N=%d D=%d kind=%d  forward: wave/task %.4f ms (%.3f of 8 TB/s)  wg/task %.4f ms (%.3f)  rel diff %.1e |  backward: wave/task %.4f ms (%.3f)  wg/task %.4f ms (%.3f)  %s"
          % (b, n, d, kind, min(fw["0"]), af / min(fw["0"]) / 8e9, min(fw["1"]), af / min(fw["1"]) / 8e9, de, min(bw["0"]), ab / min(bw["0"]) / 8e9, min(bw["1"]), ab / min(bw["1"]) / 8e9,
             "bitwise equal" if same_bwd else "BACKWARD DIFFERS"), flush=True)
