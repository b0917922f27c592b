"""Which activity of a co-resident kernel perturbs the rank-2 MLL sweep?    Spins one instruction class per
mode on a second stream while dkt_mll_f32 runs, and counts episodes whose alpha differs from the single-stream result."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402
from dkt_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B, N, D, C = 8192, 105, 1600, 5
z = torch.nn.functional.normalize(torch.randn(B, N, D, device=dev), dim=2).contiguous()
cls = torch.arange(C, device=dev).repeat_interleave(N // C)
y = torch.where(cls.unsqueeze(0) == torch.arange(C, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.full((C,), 0.69, device=dev) + 0.01 * torch.arange(C, device=dev)
mean = torch.zeros(C, device=dev)
noise = torch.full((C,), 0.1, device=dev)
e = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
torch.cuda.synchronize()
ref = ops.mll(e, y, sv, mean, noise)
torch.cuda.synchronize()
lib = dkt_amd._lib.load_diag()
lib.dkt_diag_spin.restype = ctypes.c_int
lib.dkt_diag_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
scratch = torch.empty(4096 * 256, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
names = {0: "bf16 conversions", 1: "bf16 MFMA", 2: "LDS b64 writes / b128 reads", 3: "f16 conversions + packed fp32", 4: "plain fp32 FMA"}
for mode in (4, 0, 3, 1, 2):
    worst = 0
    for rep in range(4):
        torch.cuda.synchronize()
        with torch.cuda.stream(s2):
            lib.dkt_diag_spin(scratch.data_ptr(), 4096, 3000, mode, torch.cuda.current_stream().cuda_stream)
        with torch.cuda.stream(s1):
            a = ops.mll(e, y, sv, mean, noise)
        torch.cuda.synchronize()
        worst = max(worst, int((a["alpha"] != ref["alpha"]).flatten(1).any(1).sum()))
    print("mll || %-32s episodes with different alpha (max of 4 tries): %d" % (names[mode], worst), flush=True)
