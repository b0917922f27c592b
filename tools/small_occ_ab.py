"""Occupancy A/B of the N <= 32 Gram kernels (dkt_gram_small.hip; twins library): workgroups per CU capped through DKT_GRAM_SMALL_LDS bytes of unused dynamic LDS, for
4 / 8 / 16 waves per task (DKT_GRAM_SMALL_WG = 1 / 2 / 3).  cfg0 shape: 8192 tasks of 19 x 2916.  (The first runs, profiles/r06/small_occ_ab*.log, also carried a
backward instance with five instead of eight row loads per chunk -- "nq5": slower at equal occupancy, removed.)  The product's backward launches with 56 KB.
python tools/small_occ_ab.py [N D [B]]"""
import importlib
import os
import sys

os.environ["DKT_TWINS"] = "1"
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops = importlib.import_module("deep-kernel-transfer_amd").ops
dev = torch.device("cuda:0")
n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (19, 2916)
b = int(sys.argv[3]) if len(sys.argv) > 3 else 8192


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


g = torch.Generator(device=dev).manual_seed(n + d)
z = torch.randn(b, n, d, device=dev, generator=g) * 0.05
w = torch.randn(b, n, n, device=dev, generator=g) * 0.01
ls = torch.tensor([1.3], device=dev)
res = {}
for rnd in range(3):
    for wg in ("1", "2", "3"):
        for lds in ("0", "33000", "41000", "54000", "81000"):          # at most 4 / 3 / 2 / 1 workgroups per CU (160 KB)
            os.environ["DKT_GRAM_SMALL_WG"], os.environ["DKT_GRAM_SMALL_LDS"] = wg, lds
            key = (wg, lds)
            res.setdefault(("fwd",) + key, []).append(timed(lambda: ops.gram(z, None, ops.KERNEL_RBF, ls)))
            res.setdefault(("bwd",) + key, []).append(timed(lambda: ops.gram_bwd(w, z)))
for k in ("DKT_GRAM_SMALL_WG", "DKT_GRAM_SMALL_LDS"):
    del os.environ[k]
os.environ["DKT_TWINS"] = "0"
res[("fwd", "product")] = [timed(lambda: ops.gram(z, None, ops.KERNEL_RBF, ls)) for _ in range(3)]
res[("bwd", "product")] = [timed(lambda: ops.gram_bwd(w, z)) for _ in range(3)]
print("%d tasks of %d x %d; ms (best of 3 x 10); columns: dynamic LDS 0 / 33 / 41 / 54 / 81 KB = at most all / 4 / 3 / 2 / 1 workgroups per CU" % (b, n, d))
for what in ("fwd", "bwd"):
    for wg in ("1", "2", "3"):
        print("%s waves/task %2s : %s" % (what, {"1": "4", "2": "8", "3": "16"}[wg], "  ".join("%.4f" % min(res[(what, wg, l)]) for l in ("0", "33000", "41000", "54000", "81000"))), flush=True)
    print("%s product library (default dispatch): %.4f" % (what, min(res[(what, "product")])), flush=True)
