// dkt_diag.hip -- measurement-only kernels (not part of the ABI in include/dkt_abi.h): the HBM read ceiling of the
// episode-slab access pattern of the Gram kernels with the arithmetic removed.
//   mode 0: one 256-thread workgroup per episode, [N rows] x 128-byte slices, stage by stage (the Gram forward pattern)
//   mode 1: same workgroups, but the episode's contiguous N*D block is read linearly
//   mode 2: like 0 with 256-byte slices
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {

template <int BKB>   // bytes per row slice
__global__ __launch_bounds__(256) void stream_slab_kernel(const float* __restrict__ Z, float* __restrict__ out, int N, int D) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* Zb = Z + (size_t)b * N * D;
    constexpr int V4R = BKB / 16;                     // float4 per row slice
    const int nst = (D * 4) / BKB;
    float acc = 0.f;
    for (int st = 0; st < nst; ++st) {
        for (int idx = tid; idx < N * V4R; idx += 256) {
            const int row = idx / V4R, c4 = idx % V4R;
            const float4 v = *reinterpret_cast<const float4*>(Zb + (size_t)row * D + st * (BKB / 4) + 4 * c4);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 123.456f) out[b] = acc;                // never true: keeps the loads alive
}

__global__ __launch_bounds__(256) void stream_linear_kernel(const float* __restrict__ Z, float* __restrict__ out, int N, int D) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float4* Zb = reinterpret_cast<const float4*>(Z + (size_t)b * N * D);
    const int n4 = N * D / 4;
    float acc = 0.f;
    for (int i = tid; i < n4; i += 256) {
        const float4 v = Zb[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[b] = acc;
}

// Read + write ceiling of the Gram-backward pattern: per 64-feature slab every row's 256-byte segment is read and a 256-byte
// segment of the output row written (no arithmetic, no LDS); `out` must hold B*N*D floats for modes 3 / 4.
typedef float dg_v4 __attribute__((ext_vector_type(4)));
// POL: bit 0 = non-temporal stores, bit 1 = non-temporal loads
template <int POL>
__global__ __launch_bounds__(448) void copy_slab_kernel(const float* __restrict__ Z, float* __restrict__ out, int N, int D) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* Zb = Z + (size_t)b * N * D;
    float* Ob = out + (size_t)b * N * D;
    const int nslab = D / 64;
    for (int sl = 0; sl < nslab; ++sl) {
        for (int idx = tid; idx < N * 16; idx += 448) {
            const int row = idx >> 4, c4 = idx & 15;
            const size_t o = (size_t)row * D + sl * 64 + 4 * c4;
            const dg_v4* src = reinterpret_cast<const dg_v4*>(Zb + o);
            dg_v4* dst = reinterpret_cast<dg_v4*>(Ob + o);
            dg_v4 v;
            if constexpr (POL & 2) v = __builtin_nontemporal_load(src);
            else v = *src;
            if constexpr (POL & 1) __builtin_nontemporal_store(v, dst);
            else *dst = v;
        }
    }
}

template <int POL>
__global__ __launch_bounds__(256) void copy_linear_kernel(const float* __restrict__ Z, float* __restrict__ out, int N, int D) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const dg_v4* Zb = reinterpret_cast<const dg_v4*>(Z + (size_t)b * N * D);
    dg_v4* Ob = reinterpret_cast<dg_v4*>(out + (size_t)b * N * D);
    const int n4 = N * D / 4;
    for (int i = tid; i < n4; i += 256) {
        dg_v4 v;
        if constexpr (POL & 2) v = __builtin_nontemporal_load(Zb + i);
        else v = Zb[i];
        if constexpr (POL & 1) __builtin_nontemporal_store(v, Ob + i);
        else Ob[i] = v;
    }
}

// read ceiling with non-temporal loads (the slab pattern of the Gram forward, 128-byte row slices)
__global__ __launch_bounds__(256) void stream_slab_nt_kernel(const float* __restrict__ Z, float* __restrict__ out, int N, int D) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* Zb = Z + (size_t)b * N * D;
    const int nst = D / 32;
    float acc = 0.f;
    for (int st = 0; st < nst; ++st) {
        for (int idx = tid; idx < N * 8; idx += 256) {
            const int row = idx >> 3, c4 = idx & 7;
            const dg_v4 v = __builtin_nontemporal_load(reinterpret_cast<const dg_v4*>(Zb + (size_t)row * D + st * 32 + 4 * c4));
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 123.456f) out[b] = acc;
}

}  // namespace

extern "C" int dkt_diag_stream_f32(const float* Z, float* out, int B, int N, int D, int mode, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL((stream_slab_kernel<128>), dim3(B), dim3(256), 0, st, Z, out, N, D);
    else if (mode == 2) hipLaunchKernelGGL((stream_slab_kernel<256>), dim3(B), dim3(256), 0, st, Z, out, N, D);
    else if (mode == 3) hipLaunchKernelGGL(copy_slab_kernel<0>, dim3(B), dim3(448), 0, st, Z, out, N, D);
    else if (mode == 4) hipLaunchKernelGGL(copy_linear_kernel<0>, dim3(B), dim3(256), 0, st, Z, out, N, D);
    else if (mode == 5) hipLaunchKernelGGL(copy_slab_kernel<1>, dim3(B), dim3(448), 0, st, Z, out, N, D);
    else if (mode == 6) hipLaunchKernelGGL(copy_slab_kernel<2>, dim3(B), dim3(448), 0, st, Z, out, N, D);
    else if (mode == 7) hipLaunchKernelGGL(copy_slab_kernel<3>, dim3(B), dim3(448), 0, st, Z, out, N, D);
    else if (mode == 8) hipLaunchKernelGGL(copy_linear_kernel<3>, dim3(B), dim3(256), 0, st, Z, out, N, D);
    else if (mode == 9) hipLaunchKernelGGL(stream_slab_nt_kernel, dim3(B), dim3(256), 0, st, Z, out, N, D);
    else hipLaunchKernelGGL(stream_linear_kernel, dim3(B), dim3(256), 0, st, Z, out, N, D);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---- co-residency probes (tools/coresident_probe.py): one instruction class per mode, spun for `iters` rounds ----
// Used to find out which activity of a neighbouring kernel perturbs the rank-2 MLL sweep.
typedef __bf16 dg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float dg_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void spin_kernel(float* out, int iters, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[12288];        // 48 KB, like a Gram stage buffer
    const int tid = threadIdx.x;
    float x = 1.0f + tid * 1e-3f, y = 0.5f + tid * 1e-4f, accs = 0.f;
    dg_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 12288; i += 256) lds[i] = x;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (mode == 0) {                 // fp32 -> bf16 conversions (v_cvt_pk_bf16_f32) + shifts / subtracts
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const __bf16 h = (__bf16)x;
                x = (x - (float)h) * 1.0009765625f + y;
                accs += (float)h;
            }
        } else if (mode == 1) {          // bf16 MFMA only
            dg_bf16x8 a, b;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(x + e); b[e] = (__bf16)(y - e); }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        } else if (mode == 2) {          // LDS traffic: b64 writes + b128 reads
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int o = ((tid * 4 + 1024 * k + 16 * it) % 12284) & ~3;
                *reinterpret_cast<float2*>(lds + o) = make_float2(x, y);
                const float4 v = *reinterpret_cast<const float4*>(lds + ((o + 2048) % 12284 & ~3));
                x = v.x * 0.5f + 0.25f;
                y = v.w * 0.5f + 0.125f;
            }
            __syncthreads();
        } else if (mode == 3) {          // fp32 -> f16 conversions (v_cvt_pk_f16_f32, v_cvt_f32_f16) + packed fp32 ops
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const _Float16 h = (_Float16)(x * 4096.f);
                x = (x * 4096.f - (float)h) + y;
                accs += (float)h;
            }
        } else {                         // plain fp32 VALU
#pragma unroll
            for (int k = 0; k < 32; ++k) x = __builtin_fmaf(x, 0.999f, y);
        }
    }
    out[blockIdx.x * 256 + tid] = x + y + accs + acc[0] + acc[1] + acc[2] + acc[3] + lds[tid];
}

extern "C" int dkt_diag_spin(float* out, int nblocks, int iters, int mode, void* stream) {
    hipLaunchKernelGGL(spin_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, out, iters, mode);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---- lane primitives the MFMA marginal-likelihood kernel relies on (tests/test_gpu_parity.py::test_lane_primitives) ----
// One wave.  in: [4][64] an accumulator-layout tile X (register q, lane l) and [4][64] a tile Y.
// out[0:64]    = DPP row_newbcast:3 of X register 0
// out[64:320]  = the four registers v_permlane32_swap / v_permlane16_swap spread X register 1 into
// out[320:576] = X^T Y through four v_mfma_f32_16x16x4_f32 with X's registers as A and Y's as B operands (accumulator layout)
// out[576:640] = v_fmac_f32_dpp acc += row_newbcast:5(acc) * t  with acc = X register 2, t = Y register 0
__global__ __launch_bounds__(64) void lane_primitives_kernel(const float* in, float* out) {
    const int l = threadIdx.x;
    float x[4], y[4];
    for (int q = 0; q < 4; ++q) { x[q] = in[q * 64 + l]; y[q] = in[256 + q * 64 + l]; }
    out[l] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[0]), 0x150 + 3, 0xf, 0xf, false));
    const unsigned u = __float_as_uint(x[1]);
    auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    auto lo = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(h[1], h[1], false, false);
    out[64 + l] = __uint_as_float(lo[0]);
    out[128 + l] = __uint_as_float(lo[1]);
    out[192 + l] = __uint_as_float(hi[0]);
    out[256 + l] = __uint_as_float(hi[1]);
    dg_f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < 4; ++q) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q], y[q], c, 0, 0, 0);
    for (int q = 0; q < 4; ++q) out[320 + q * 64 + l] = c[q];
    float acc = x[2];
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(y[0]));
    out[576 + l] = acc;
}

extern "C" int dkt_diag_lane_primitives(const float* in, float* out, void* stream) {
    hipLaunchKernelGGL(lane_primitives_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---- issue-rate microbenchmarks of the instruction forms the MFMA marginal-likelihood sweep uses (tools/ubench_valu.py) ----
// One wave per workgroup; out[block * 8 + v] = s_memtime ticks per instruction of variant v.
#define DG_REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <bool BIGREG>
__global__ __launch_bounds__(768) void valu_ubench_kernel(float* out, int iters) {
    if constexpr (BIGREG) asm volatile("v_mov_b32 v167, 0" ::: "v167");           // 168 VGPRs: three waves fill a SIMD's register file
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    float t = 1e-6f;
    unsigned long long t0, t1;
    float res[8];
    // 0: plain v_fmac_f32
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define DG_X(k) asm volatile("v_fmac_f32 %0, %0, %1" : "+v"(a[k]) : "v"(t));
        DG_REP16(DG_X)
#undef DG_X
    }
    t1 = __builtin_amdgcn_s_memtime();
    res[0] = (float)(t1 - t0) / (16.f * iters);
    // 1: v_fmac_f32_dpp row_newbcast
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define DG_X(k) asm volatile("v_fmac_f32_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(t));
        DG_REP16(DG_X)
#undef DG_X
    }
    t1 = __builtin_amdgcn_s_memtime();
    res[1] = (float)(t1 - t0) / (16.f * iters);
    // 2: v_fmac_f32_dpp quad_perm (classic DPP)
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define DG_X(k) asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[k]) : "v"(t));
        DG_REP16(DG_X)
#undef DG_X
    }
    t1 = __builtin_amdgcn_s_memtime();
    res[2] = (float)(t1 - t0) / (16.f * iters);
    // 3: v_readlane_b32 + v_fmac_f32 with the SGPR operand (2 instructions per update)
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define DG_X(k) asm volatile("v_readlane_b32 s40, %0, 3\n\tv_fmac_f32 %0, s40, %1" : "+v"(a[k]) : "v"(t) : "s40");
        DG_REP16(DG_X)
#undef DG_X
    }
    t1 = __builtin_amdgcn_s_memtime();
    res[3] = (float)(t1 - t0) / (32.f * iters);
    // 4: v_mov_b32_dpp row_newbcast
    float b[16];
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define DG_X(k) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(b[k]) : "v"(a[k]));
        DG_REP16(DG_X)
#undef DG_X
    }
    t1 = __builtin_amdgcn_s_memtime();
    res[4] = (float)(t1 - t0) / (16.f * iters);
    // 5: v_rsq_f32
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define DG_X(k) asm volatile("v_rsq_f32 %0, %1" : "=v"(b[k]) : "v"(a[k]));
        DG_REP16(DG_X)
#undef DG_X
    }
    t1 = __builtin_amdgcn_s_memtime();
    res[5] = (float)(t1 - t0) / (16.f * iters);
    // 6: dependent chain of plain v_fmac_f32 (latency)
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define DG_X(k) asm volatile("v_fmac_f32 %0, %0, %1" : "+v"(a[0]) : "v"(t));
        DG_REP16(DG_X)
#undef DG_X
    }
    t1 = __builtin_amdgcn_s_memtime();
    res[6] = (float)(t1 - t0) / (16.f * iters);
    // 7: v_mfma_f32_16x16x4_f32, four independent accumulators
    dg_f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], a[3], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4], a[5], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[6], a[7], c3, 0, 0, 0);
        }
    }
    t1 = __builtin_amdgcn_s_memtime();
    res[7] = (float)(t1 - t0) / (16.f * iters);
    float s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 16; ++i) s += a[i] + b[i];
    if ((threadIdx.x & 63) == 0)
        for (int v = 0; v < 8; ++v) out[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 8 + v] = res[v];
    if (s == 123.456f) out[0] = s;
}

extern "C" int dkt_diag_valu_ubench(float* out, int nblocks, int waves_per_block, int iters, void* stream) {
    if (iters < 0) hipLaunchKernelGGL(valu_ubench_kernel<true>, dim3(nblocks), dim3(64 * waves_per_block), 0, (hipStream_t)stream, out, -iters);
    else hipLaunchKernelGGL(valu_ubench_kernel<false>, dim3(nblocks), dim3(64 * waves_per_block), 0, (hipStream_t)stream, out, iters);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---- the diagonal-tile sweep of dkt_mll_mfma.hip in isolation, at 1 .. 3 waves per SIMD (tools/ubench_valu.py) ----
// MODE 0: v_fmac_f32_dpp (fused);  MODE 1: v_mov_b32_dpp + v_fmac_f32;  MODE 2: fused, no s_nop between the pieces
template <int MODE, int P, int I>
__device__ __forceinline__ void dg_rows(float (&x)[16], const float t) {
    if constexpr (I < 16) {
        if constexpr (MODE == 1) {
            float m;
            asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(x[I]), "n"(P));
            x[I] = __builtin_fmaf(m, t, x[I]);
        } else if constexpr (MODE == 0) {
            asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x[I]) : "v"(t), "n"(P));
        } else {
            asm volatile("v_fmac_f32_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x[I]) : "v"(t), "n"(P));
        }
        dg_rows<MODE, P, I + 1>(x, t);
    }
}
template <int MODE, int P>
__device__ __forceinline__ void dg_sweep(float (&x)[16], float& dv, const int c) {
    if constexpr (P < 16) {
        float xp;
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(xp) : "v"(x[P]), "n"(P));
        const float d = 1.0f - xp;
        const bool eq = c == P;
        dv = eq ? d : dv;
        const float rs = __builtin_amdgcn_rsqf(d);
        const float t = x[P] * (rs * rs);
        x[P] = eq ? rs : x[P] * rs;
        dg_rows<MODE, P, P + 1>(x, t);
        dg_sweep<MODE, P + 1>(x, dv, c);
    }
}
template <int MODE>
__global__ __launch_bounds__(768) void sweep_ubench_kernel(float* out, int iters) {
    float x[16], dv = 0.f;
    for (int i = 0; i < 16; ++i) x[i] = ((threadIdx.x & 15) == i) ? 0.5f : 0.01f * (i + 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        dg_sweep<MODE, 0>(x, dv, threadIdx.x & 15);
        for (int i = 0; i < 16; ++i) x[i] = x[i] * 1e-3f + (((threadIdx.x & 15) == i) ? 0.5f : 0.01f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = dv;
    for (int i = 0; i < 16; ++i) s += x[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (float)(t1 - t0) / iters;
    if (s == 123.456f) out[0] = s;
}
extern "C" int dkt_diag_sweep_ubench(float* out, int nblocks, int waves_per_block, int iters, int mode, void* stream) {
    const dim3 g(nblocks), b(64 * waves_per_block);
    if (mode == 0) hipLaunchKernelGGL(sweep_ubench_kernel<0>, g, b, 0, (hipStream_t)stream, out, iters);
    else if (mode == 1) hipLaunchKernelGGL(sweep_ubench_kernel<1>, g, b, 0, (hipStream_t)stream, out, iters);
    else hipLaunchKernelGGL(sweep_ubench_kernel<2>, g, b, 0, (hipStream_t)stream, out, iters);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---- do fp32 MFMAs and VALU instructions of DIFFERENT waves of one SIMD overlap?  (tools/ubench_valu.py) ----
// 8 waves per workgroup = 2 per SIMD.  role 0: waves 0-3 run an MFMA stream, waves 4-7 a v_fmac stream, at the same time;
// role 1: everybody MFMA;  role 2: everybody v_fmac;  role 3: waves 0-3 bf16 MFMA (XDL pipe), waves 4-7 v_fmac.
// out[wave] = ticks per instruction of that wave's stream.
__global__ __launch_bounds__(512) void overlap_ubench_kernel(float* out, int iters, int role) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_wave = role == 1 || ((role == 0 || role == 3) && wave < 4);
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    float t = 1e-6f;
    dg_f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mfma_wave && role != 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], a[3], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[4], a[5], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[6], a[7], c3, 0, 0, 0);
            }
        }
    } else if (mfma_wave) {
        dg_bf16x8 x, y;
        for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(a[e]); y[e] = (__bf16)(a[8 + e]); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c3, 0, 0, 0);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#define DG_X(k) asm volatile("v_fmac_f32 %0, %0, %1" : "+v"(a[k]) : "v"(t));
            DG_REP16(DG_X)
#undef DG_X
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 16; ++i) s += a[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (float)(t1 - t0) / (16.f * iters);
    if (s == 123.456f) out[0] = s;
}
extern "C" int dkt_diag_overlap_ubench(float* out, int nblocks, int iters, int role, void* stream) {
    hipLaunchKernelGGL(overlap_ubench_kernel, dim3(nblocks), dim3(512), 0, (hipStream_t)stream, out, iters, role);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// ---- the f16-split tile primitives of dkt_h2_tiles.h in isolation (tests/test_gpu_parity.py::test_h2_tile_primitives) ----
// in: X[16][16], Y[16][16] row-major, sx, sy.  out (row-major 16 x 16 each): 0: join(split(X, sx)) = sx X;
// 1: split(X, sx)^T split(Y, sy) (three plane products) = sx sy X^T Y;  2: join(neg_transpose(split(X, sx))) = -sx X^T;
// 3: the h plane of split(X, sx) as fp32;  4: the m plane.
#include "dkt_h2_tiles.h"
namespace {
__global__ __launch_bounds__(64) void h2_primitives_kernel(const float* in, float* out, float sx, float sy) {
    using namespace dkt_mfma;
    const int l = threadIdx.x, g4 = (l >> 2) & 12, c = l & 15;
    f32x4 X, Y;
    for (int q = 0; q < 4; ++q) { X[q] = in[(g4 + q) * 16 + c]; Y[q] = in[256 + (g4 + q) * 16 + c]; }
    sx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sx)));
    sy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sy)));
    const f32x4 xs = split_h2(X, sx), ys = split_h2(Y, sy);
    const f32x4 j = join_h2(xs);
    const f32x4 p = xtyh0(xs, ys);
    h4 negI;
    for (int q = 0; q < 4; ++q) negI[q] = (g4 + q == c) ? (_Float16)-1.0f : (_Float16)0.0f;
    const f32x4 t = join_h2(neg_transpose_h2(xs, negI));
    const Sp sp = as_sp(xs);
    for (int q = 0; q < 4; ++q) {
        out[0 * 256 + (g4 + q) * 16 + c] = j[q];
        out[1 * 256 + (g4 + q) * 16 + c] = p[q];
        out[2 * 256 + (g4 + q) * 16 + c] = t[q];
        out[3 * 256 + (g4 + q) * 16 + c] = (float)sp.h[q];
        out[4 * 256 + (g4 + q) * 16 + c] = (float)sp.m[q];
    }
}

// issue rates of the f16 matrix instructions next to VALU work (tools/ubench_valu.py): role 0: all 8 waves v_mfma_f32_16x16x16_f16
// (4 chains); 1: all waves v_mfma_f32_16x16x32_f16; 2: waves 0-3 16x16x16 f16, waves 4-7 v_fmac_f32; 3: every wave alternates
// one 16x16x16 f16 MFMA with 4 v_fmac_f32; 4: every wave alternates one fp32 16x16x4 MFMA with 4 v_fmac_f32; 5: every wave alternates one
// 16x16x16 f16 MFMA with 4 v_fma_mixlo_f16.  out[wave] = ticks per MFMA (VALU-only waves: per v_fmac).
template <int ROLE>
__global__ __launch_bounds__(512) void h2_ubench_kernel(float* out, int iters) {
    using namespace dkt_mfma;
    const int wave = threadIdx.x >> 6;
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    float t = 1e-6f;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    h4 x, y;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 x8, y8;
    for (int e = 0; e < 4; ++e) { x[e] = (_Float16)a[e]; y[e] = (_Float16)a[4 + e]; }
    for (int e = 0; e < 8; ++e) { x8[e] = (_Float16)a[e]; y8[e] = (_Float16)a[8 + e]; }
    __syncthreads();
    float per = 16.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (ROLE == 0 || (ROLE == 2 && wave < 4)) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(x, y, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(x, y, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(x, y, c3, 0, 0, 0);
            }
        }
    } else if (ROLE == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, c3, 0, 0, 0);
            }
        }
    } else if (ROLE == 2) {
        for (int it = 0; it < iters; ++it) {
#define DG_X(k) asm volatile("v_fmac_f32 %0, %0, %1" : "+v"(a[k]) : "v"(t));
            DG_REP16(DG_X)
#undef DG_X
        }
    } else {
        // one matrix instruction, then four VALU instructions, four times per iteration: per = groups per iteration
        per = 4.f;
        for (int it = 0; it < iters; ++it) {
#define DG_M(cc)                                                                                    \
    if (ROLE == 4) cc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], cc, 0, 0, 0);              \
    else cc = __builtin_amdgcn_mfma_f32_16x16x16f16(x, y, cc, 0, 0, 0);                             \
    __builtin_amdgcn_sched_barrier(0);
#define DG_V(k)                                                                                     \
    if (ROLE == 5) asm volatile("v_fma_mixlo_f16 %0, %1, %1, 0" : "+v"(a[k]) : "v"(t));            \
    else asm volatile("v_fmac_f32 %0, %0, %1" : "+v"(a[k]) : "v"(t));
            DG_M(c0) DG_V(2) DG_V(3) DG_V(4) DG_V(5) __builtin_amdgcn_sched_barrier(0);
            DG_M(c1) DG_V(6) DG_V(7) DG_V(8) DG_V(9) __builtin_amdgcn_sched_barrier(0);
            DG_M(c2) DG_V(10) DG_V(11) DG_V(12) DG_V(13) __builtin_amdgcn_sched_barrier(0);
            DG_M(c3) DG_V(14) DG_V(15) DG_V(2) DG_V(3) __builtin_amdgcn_sched_barrier(0);
#undef DG_M
#undef DG_V
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 16; ++i) s += a[i];
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = (float)(t1 - t0) / (per * iters);
    if (s == 123.456f) out[0] = s;
}
}  // namespace
extern "C" int dkt_diag_h2_primitives(const float* in, float* out, float sx, float sy, void* stream) {
    hipLaunchKernelGGL(h2_primitives_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out, sx, sy);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
extern "C" int dkt_diag_h2_ubench(float* out, int nblocks, int iters, int role, void* stream) {
    const dim3 g(nblocks), b(512);
    hipStream_t st = (hipStream_t)stream;
    switch (role) {
        case 0: hipLaunchKernelGGL(h2_ubench_kernel<0>, g, b, 0, st, out, iters); break;
        case 1: hipLaunchKernelGGL(h2_ubench_kernel<1>, g, b, 0, st, out, iters); break;
        case 2: hipLaunchKernelGGL(h2_ubench_kernel<2>, g, b, 0, st, out, iters); break;
        case 3: hipLaunchKernelGGL(h2_ubench_kernel<3>, g, b, 0, st, out, iters); break;
        case 4: hipLaunchKernelGGL(h2_ubench_kernel<4>, g, b, 0, st, out, iters); break;
        default: hipLaunchKernelGGL(h2_ubench_kernel<5>, g, b, 0, st, out, iters); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
