"""Occupancy A/B of the episode-resident unit-row Gram kernels (dkt_gram_ep.hip; twins library): DKT_PAD_GRAM_EP / DKT_PAD_GRAM_EP_BWD bytes of untouched dynamic LDS per
launch cap the workgroups per CU.  Forward <7,...>: 48 KB static LDS + 158 registers = three workgroups of four waves; backward <7,2,2,1>: 72 KB = two of seven waves.
python tools/gram_ep_occ_ab.py"""
import os
import sys

os.environ["DKT_TWINS"] = "force"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / reps


g = torch.Generator(device=dev).manual_seed(1)
SHAPES = [(8192, 105, 1600), (8192, 85, 512), (2048, 105, 1600), (8192, 128, 1600)] if len(sys.argv) < 2 else [(8192, 85, 512), (8192, 105, 512), (8192, 96, 256), (2048, 85, 512), (8192, 85, 1024)]
for (b, n, d) in SHAPES:
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=dev), dim=2).contiguous()
    w = torch.randn(b, n, n, generator=g, device=dev)
    w = (w + w.transpose(1, 2)).contiguous()
    fw, bw = {}, {}
    pads_f = ("0", "20000", "40000", "70000")          # forward: 48 KB static: 3 / 2 (68 KB) / 1 (88 KB) / 1 workgroups per CU
    pads_b = ("0", "10000", "20000") if len(sys.argv) < 2 else ("0", "6000", "12000", "20000", "32000", "60000")     # backward <7,2,2,1>: 72 KB static: 2 / 1 / 1; the one-image instance below D = 1024: ~50 KB
    for rnd in range(3):
        for p in pads_f:
            os.environ["DKT_PAD_GRAM_EP"] = p
            fw.setdefault(p, []).append(timed(lambda: ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)))
        for p in pads_b:
            os.environ["DKT_PAD_GRAM_EP_BWD"] = p
            bw.setdefault(p, []).append(timed(lambda: ops.gram_bwd(w, z, None, unit_rows=True, w_symmetric=True)))
    del os.environ["DKT_PAD_GRAM_EP"], os.environ["DKT_PAD_GRAM_EP_BWD"]
    print("B=%d N=%d D=%d  forward, pad %s: %s ms | backward, pad %s: %s ms" % (b, n, d, "/".join(pads_f), "  ".join("%.4f" % min(fw[p]) for p in pads_f),
                                                                            "/".join(pads_b), "  ".join("%.4f" % min(bw[p]) for p in pads_b)), flush=True)
