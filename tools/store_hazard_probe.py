"""Hardware probe for the hazard behind bstore4 (dkt_mfma_tiles.h): a 16-byte buffer store with an SGPR soffset followed AT ONCE by a VALU write of one of
its data registers.  hipcc emits that pair without a wait state; this compiles a kernel that contains it (checked with the scanner of
_lib.unprotected_wide_buffer_stores), runs it on the GPU and compares what arrived in memory with the values the program order defines.
    python tools/store_hazard_probe.py"""
import ctypes
import importlib
import os
import subprocess
import sys
import tempfile
import textwrap

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("deep-kernel-transfer_amd")._lib
SRC = textwrap.dedent("""
    #include <hip/hip_runtime.h>
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    template <bool SGPR_SOFFSET>
    __global__ void probe(float* p, int nbytes, int so) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, nbytes, 0x00020000);
        const int t = blockIdx.x * blockDim.x + threadIdx.x;
        float a = (float)(t & 1023), b = a + 1.f, c = a + 2.f, d = a + 3.f;
    #pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
            if (SGPR_SOFFSET) __builtin_amdgcn_raw_buffer_store_b128(v, r, t * 16, so * i, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(v, r, t * 16 + so * i, 0, 0);
            a = a + b; b = b + c; c = c + d; d = d + a;          // exact in fp32 for these magnitudes: the host reference is unambiguous
        }
    }
    extern "C" void run(float* p, int nbytes, int so, int blocks, int sgpr, void* stream) {
        if (sgpr) hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, nbytes, so);
        else hipLaunchKernelGGL(probe<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, nbytes, so);
    }""")
td = tempfile.mkdtemp()
open(os.path.join(td, "p.hip"), "w").write(SRC)
so_path = os.path.join(td, "libprobe.so")
subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(td, "p.hip"), "-o", so_path], check=True)
pairs = L.unprotected_wide_buffer_stores(so_path)
print("store / overwrite pairs the compiler emitted without a wait state: %d" % len(pairs))
for k, s, v in pairs[:4]:
    print("   %s:  %s  ->  %s" % (k[:40], s, v))
lib = ctypes.CDLL(so_path)
lib.run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
blocks = 4096
nthr = blocks * 256
so = nthr * 16
t = torch.arange(nthr, dtype=torch.float64) % 1024
a, b, c, d = t.clone(), t + 1, t + 2, t + 3
ref = torch.empty(8, nthr, 4, dtype=torch.float64)
for i in range(8):
    ref[i] = torch.stack([a, b, c, d], 1)
    a = a + b; b = b + c; c = c + d; d = d + a
ref = ref.float().to(dev)
for sgpr in (1, 0):
    bad_runs, worst = 0, 0
    for rep in range(20):
        buf = torch.full((8, nthr, 4), -1.0, device=dev)
        lib.run(buf.data_ptr(), buf.numel() * 4, so, blocks, sgpr, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        nbad = int((buf != ref).sum().item())
        bad_runs += nbad > 0
        worst = max(worst, nbad)
    print("%s: %d of 20 runs stored wrong data (worst run: %d of %d floats)" % ("SGPR soffset (unprotected)" if sgpr else "offset in the VGPR, soffset 0 ", bad_runs, worst, ref.numel()))
