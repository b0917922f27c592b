"""Time dkt_gram_f32 at cfg2 for the library selected by DKT_AMD_LIB (kernel-variant experiments)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402,F401
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
z = torch.nn.functional.normalize(torch.randn(2048, 105, 1600, device=dev), dim=2).contiguous()
for rep in range(3):
    for _ in range(5):
        ops.gram(z)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        ops.gram(z)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 50
    print("gram cfg2 B=2048: %.4f ms  %.0f GB/s" % (ms, 4.0 * 2048 * (105 * 1600 + 105 * 105) / ms / 1e6), flush=True)
