"""Phase breakdown of the split Gram forward kernel from the DKT_EXP_CLOCKS measurement build (wave 0 of every workgroup
accumulates clock64() deltas per phase and dumps them over E[b][0][0:8])."""
import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
z = torch.nn.functional.normalize(torch.randn(2048, 105, 1600, device=dev), dim=2).contiguous()
names = ["issue loads", "MFMA phase (+frag reads)", "barrier 1", "load wait (vmcnt 0)", "split + LDS store", "barrier 2", "whole loop", "epilogue"]
for var in ("11", "12"):
    os.environ["DKT_GRAM_SPLIT_VAR"] = var
    for _ in range(3):
        e = ops.gram(z)
    torch.cuda.synchronize()
    s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    e = ops.gram(z)
    t.record()
    torch.cuda.synchronize()
    c = e[:, 0, :8].double().mean(0).cpu().tolist()
    tot = sum(c[:6])
    print("var %s  kernel %.4f ms; per-workgroup clock ticks (mean over %d workgroups):" % (var, s.elapsed_time(t), z.shape[0]))
    for n, v in zip(names, c):
        print("   %-28s %10.0f  %5.1f%%" % (n, v, 100.0 * v / tot))
