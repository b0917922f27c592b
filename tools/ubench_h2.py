"""Issue rates of the f16 matrix instructions the f16-split marginal-likelihood kernel uses, alone and next to VALU work
(s_memtime ticks per instruction, two waves per SIMD).  Measurement tooling; kernels in csrc/dkt_diag.hip."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402

lib = dkt_amd._lib.load_diag()
lib.dkt_diag_h2_ubench.restype = ctypes.c_int
lib.dkt_diag_h2_ubench.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
names = {0: "all waves: v_mfma_f32_16x16x16_f16 (4 chains)", 1: "all waves: v_mfma_f32_16x16x32_f16 (4 chains)",
         2: "waves 0-3 16x16x16 f16 MFMA | waves 4-7 v_fmac_f32", 3: "every wave: 1 f16 16x16x16 MFMA + 4 v_fmac_f32, per group",
         4: "every wave: 1 fp32 16x16x4 MFMA + 4 v_fmac_f32, per group", 5: "every wave: 1 f16 16x16x16 MFMA + 4 v_fma_mixlo_f16, per group"}
for role in range(6):
    out = torch.zeros(256 * 8, device="cuda")
    for _ in range(2):
        lib.dkt_diag_h2_ubench(out.data_ptr(), 256, 1000, role, None)
    torch.cuda.synchronize()
    o = out.view(256, 8).mean(0).cpu().numpy()
    print("  %-62s %.2f | %.2f" % (names[role], o[:4].mean(), o[4:].mean()))
