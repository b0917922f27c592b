"""(The DKT_GRAM_SMALL_PF instances were removed after this measurement -- commit 06c82cc has them; the script still sweeps the cap for the shipped depth.)
A/B of the N <= 32 Gram forward's prefetch depth (dkt_gram_small.hip; twins library): DKT_GRAM_SMALL_PF=8 (eight 16-feature steps in flight per wave instead of four)
under the workgroups-per-CU cap DKT_GRAM_SMALL_LDS.   python tools/small_pf_ab.py"""
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


for (b, n, d, kind) in [(8192, 19, 2916, ops.KERNEL_RBF), (8192, 19, 2916, ops.KERNEL_LINEAR), (8192, 25, 2048, ops.KERNEL_LINEAR), (1024, 19, 2916, ops.KERNEL_RBF)]:
    g = torch.Generator(device=dev).manual_seed(n + d)
    z = torch.randn(b, n, d, device=dev, generator=g) * 0.05
    ls = torch.tensor([1.3], device=dev)
    res, outs = {}, {}
    for rnd in range(3):
        for pf in ("4", "8"):
            for lds in ("0", "33000", "41000", "54000", "81000"):
                os.environ["DKT_GRAM_SMALL_PF"], os.environ["DKT_GRAM_SMALL_LDS"] = pf, lds
                ms, e = timed(lambda: ops.gram(z, None, kind, ls if kind != ops.KERNEL_LINEAR else None))
                res.setdefault((pf, lds), []).append(ms)
                outs[pf] = e
    del os.environ["DKT_GRAM_SMALL_PF"], os.environ["DKT_GRAM_SMALL_LDS"]
    print("B=%d N=%d D=%d kind=%d  (columns: at most all / 4 / 3 / 2 / 1 workgroups per CU)  %s" % (b, n, d, kind, "bitwise equal" if torch.equal(outs["4"], outs["8"]) else "DIFFER"))
    for pf in ("4", "8"):
        print("   %s steps in flight: %s" % (pf, "  ".join("%.4f" % min(res[(pf, l)]) for l in ("0", "33000", "41000", "54000", "81000"))), flush=True)
