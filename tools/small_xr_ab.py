"""(The "c" column = DKT_GRAM_SMALL_COAL, coalesced loads + a lane transpose, was removed after this measurement -- commit 06c82cc has it; the switch is ignored now.)
A/B of the N <= 32 Gram forward with the rows beyond sixteen on the VALU (dkt_gram_small.hip, template parameter XR; 17 <= N <= 20: the QMUL head's 19 frames) against
the three-MFMA-tile form (twins library, DKT_GRAM_SMALL_XR=0); errors of both against float64.   python tools/small_xr_ab.py"""
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


for (b, n, d, kind) in [(8192, 19, 2916, ops.KERNEL_RBF), (8192, 19, 2916, ops.KERNEL_LINEAR), (8192, 25, 2916, ops.KERNEL_RBF), (8192, 25, 2048, ops.KERNEL_LINEAR),
                        (8192, 19, 512, ops.KERNEL_RBF), (8192, 19, 64, ops.KERNEL_LINEAR), (1024, 19, 2916, ops.KERNEL_RBF), (64, 19, 2916, ops.KERNEL_RBF)]:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.randn(b, n, d, device=dev, generator=g) * 0.05
    ls = torch.tensor([1.3], device=dev)
    fw, outs = {}, {}
    for rnd in range(3):
        for v in ("0", "1", "c"):
            os.environ["DKT_GRAM_SMALL_XR"] = "1" if v == "c" else v                      # "c": coalesced loads + the lane transpose (DKT_GRAM_SMALL_COAL=1)
            os.environ["DKT_GRAM_SMALL_COAL"] = "1" if v == "c" else "0"
            if v == "c" and kind == ops.KERNEL_LINEAR:
                os.environ["DKT_GRAM_SMALL_XR"] = "0"
            ms, e = timed(lambda: ops.gram(z, None, kind, ls if kind != ops.KERNEL_LINEAR else None))
            fw.setdefault(v, []).append(ms)
            outs[v] = e
    del os.environ["DKT_GRAM_SMALL_XR"], os.environ["DKT_GRAM_SMALL_COAL"]
    zd = z[:256].double()
    ref = zd @ zd.transpose(1, 2)
    if kind == ops.KERNEL_RBF:
        dg = torch.diagonal(ref, dim1=1, dim2=2)
        ref = torch.exp(-0.5 * (dg.unsqueeze(2) + dg.unsqueeze(1) - 2 * ref).clamp_min(0) / 1.3 ** 2)
    err = {v: ((outs[v][:256].double() - ref).abs().max() / ref.abs().max()).item() for v in ("0", "1", "c")}
    sym = bool(torch.equal(outs["1"], outs["1"].transpose(1, 2)))
    af = b * (n * d + n * n) * 4
    print("B=%d N=%d D=%d kind=%d  MFMA tiles %.4f ms (%.3f of 8 TB/s, err %.1e)   extra rows on the VALU %.4f ms (%.3f, err %.1e, %s)   coalesced loads + lane transpose %.4f ms (%.3f, err %.1e, %s)"
          % (b, n, d, kind, min(fw["0"]), af / min(fw["0"]) / 8e9, err["0"], min(fw["1"]), af / min(fw["1"]) / 8e9, err["1"], "symmetric" if sym else "NOT SYMMETRIC",
             min(fw["c"]), af / min(fw["c"]) / 8e9, err["c"], "bitwise = its form without" if torch.equal(outs["c"], outs["1" if kind == ops.KERNEL_RBF and 16 < n <= 19 else "0"]) else "differs"), flush=True)
