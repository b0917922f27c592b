"""Issue rates (s_memtime ticks per instruction, one wave per SIMD) of the instruction forms the MFMA marginal-likelihood
sweep is built from: plain / DPP-fused FMA, readlane + FMA, DPP move, rsq, a dependent FMA chain, fp32 MFMA.  Measurement tooling."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402

lib = dkt_amd._lib.load_diag()
lib.dkt_diag_valu_ubench.restype = ctypes.c_int
lib.dkt_diag_valu_ubench.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
names = ["v_fmac_f32", "v_fmac_f32_dpp row_newbcast", "v_fmac_f32_dpp quad_perm", "v_readlane + v_fmac (per instr)", "v_mov_b32_dpp row_newbcast",
         "v_rsq_f32", "v_fmac_f32 dependent chain", "v_mfma_f32_16x16x4_f32 (4 chains)"]
for nb, wpb, it in ((1, 1, 500), (256, 8, 500), (256, 12, 500), (256, 8, -500), (256, 10, -500), (256, 12, -500)):      # iters < 0: 168 VGPRs per wave
    out = torch.zeros(nb * wpb * 8, device="cuda")
    for _ in range(2):
        lib.dkt_diag_valu_ubench(out.data_ptr(), nb, wpb, it, None)
    torch.cuda.synchronize()
    r = out.view(nb * wpb, 8).mean(0).cpu().numpy()
    print("blocks = %d, waves per block = %d%s" % (nb, wpb, ", 168 VGPRs" if it < 0 else ""))
    for n, v in zip(names, r):
        print("  %-36s %.2f ticks / instruction per wave" % (n, v))

lib.dkt_diag_sweep_ubench.restype = ctypes.c_int
lib.dkt_diag_sweep_ubench.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
print("one 16 x 16 diagonal-tile sweep (ticks per sweep per wave):")
for mode, name in ((0, "v_fmac_f32_dpp, s_nop 1 per update"), (2, "v_fmac_f32_dpp, no s_nop"), (1, "v_mov_b32_dpp + v_fmac_f32")):
    row = []
    for nb, wpb in ((1, 1), (256, 4), (256, 8), (256, 10), (256, 12)):
        out = torch.zeros(nb * wpb, device="cuda")
        for _ in range(2):
            lib.dkt_diag_sweep_ubench(out.data_ptr(), nb, wpb, 200, mode, None)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        row.append("%d w/blk: mean %.0f max %.0f" % (wpb, o.mean(), o.max()))
    print("  %-38s %s" % (name, " | ".join(row)))

lib.dkt_diag_overlap_ubench.restype = ctypes.c_int
lib.dkt_diag_overlap_ubench.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
print("two waves per SIMD (ticks per instruction; waves 0-3 | waves 4-7):")
for role, name in ((1, "all waves: fp32 MFMA 16x16x4"), (2, "all waves: v_fmac_f32"), (0, "waves 0-3 fp32 MFMA, waves 4-7 v_fmac_f32"),
                   (3, "waves 0-3 bf16 MFMA 16x16x32, waves 4-7 v_fmac_f32")):
    out = torch.zeros(256 * 8, device="cuda")
    for _ in range(2):
        lib.dkt_diag_overlap_ubench(out.data_ptr(), 256, 1000, role, None)
    torch.cuda.synchronize()
    o = out.view(256, 8).mean(0).cpu().numpy()
    print("  %-52s %.2f | %.2f" % (name, o[:4].mean(), o[4:].mean()))
