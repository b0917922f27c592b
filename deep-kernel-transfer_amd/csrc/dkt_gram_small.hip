// dkt_gram_small.hip -- Gram build and Gram backward for SMALL conditioning sets (N <= 32): one WAVE per episode / task, no LDS
// staging, no barrier.  The regression head of the reference works on 19 frames of one person (methods/DKT_regression.py:45-64,
// qmul_loader.py: Z is [19, 2916]) and a 5-way 5-shot test episode conditions on 25 support features (methods/DKT.py:224-240): a
// 64 x 64 output tile per 256-thread workgroup (dkt_gram.hip) fills a third of its MFMA tile and a tenth of the HBM roof there.
//
// Replaces ExactGPLayer.forward -> covar_module(x) (reference methods/DKT.py:375-378, methods/DKT_regression.py:126-129:
// LinearKernel / RBFKernel evaluation) and autograd through it (DKT.py:163, DKT_regression.py:56) at these sizes.
//
// Forward, E = k(Z, Z): v_mfma_f32_16x16x4_f32 takes ONE float per lane for A (lane (g, r): A[r][k = g]) and for B (B[k = g][r]), so
// the float4 a lane loads from row 16 blk + r at features 16 s + 4 g .. + 3 is four A operands AND four B operands of the episode's
// own Gram (a k-permutation: the t-th MFMA of the step contracts the features 16 s + 4 g + t) -- the operands go from the global
// load straight into the matrix instruction, coalesced as 64-byte row segments.  Exact fp32 (bitwise a k-ordered fmaf chain).  RBF /
// squared distances as in dkt_gram.hip: rows shifted by row 0 (the role of GPyTorch's mean-centring), d2 = G_ii + G_jj - 2 G_ij from
// the Gram's own diagonal (exactly 0 on the diagonal), clamped at 0.
// Backward, dZ = s (W + W^T) Z: the contraction runs over the rows j of Z, which a lane reads as float4 of row 4 kk + g at columns
// d0 + 4 c .. + 3: element e is the B operand of output tile e, whose column c therefore stands for the feature d0 + 4 c + e -- the
// four accumulators of a lane are four consecutive features of one row: one 16-byte store.  (W + W^T) sits in 2 x 8 registers.
// HBM traffic = the algorithmic bytes: Z once (forward), Z + dZ once (backward).
#include "dkt_common.h"
#include "../../include/dkt_abi.h"

namespace {

typedef __amdgpu_buffer_rsrc_t brsrc;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int SM_OOB = 0x7ffffff0;      // an offset every descriptor rejects: the load returns 0, the store is dropped

__device__ __forceinline__ brsrc sm_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 sm_load4(brsrc r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

// the value of lane I of the own 16-lane group (ds_swizzle, bit-mask mode: lane' = (lane & 0x10) | I inside each group of 32; the crossbar only, no LDS storage)
template <int I>
__device__ __forceinline__ float sm_bcast16(const float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (I << 5) | 0x10));
}

constexpr int SM_PF = 4;                 // 16-feature steps in flight

// WGT (round 5): a WORKGROUP per task instead of a wave per task -- wave w takes the feature blocks w, w + 4, ... (64 features each), so that the four waves
// of a workgroup pull 1 KB of every row at a time instead of 256 B of the rows of four different tasks; the partial Grams meet in LDS at the end.
// XR (round 6; NB = 2 and 16 < N <= 16 + XR <= 20 -- the QMUL head's 19 frames): the XR rows beyond the first sixteen fill 3 / 16 of the two tiles they add, i.e. two of
// the three MFMAs of a step multiply padding.  With XR > 0 only tile (0, 0) runs on the matrix pipe; row 16 + i is broadcast inside every 16-lane group (ds_swizzle:
// no LDS storage involved) and its products with the sixteen rows above and with the other extra rows are 2 x 4 fused multiply-adds per lane, summed over the four
// feature groups of a wave at the end (fixed order) into the same accumulator layout the epilogue reads.
template <int KIND, int NB, int WGT, int XR = 0>     // NB = 1: N <= 16, NB = 2: N <= 32; WGT = waves of a workgroup that share a task (0: a wave per task, 4, 8)
__global__ __launch_bounds__(WGT > 4 ? 64 * WGT : 256) void gram_small_kernel(const float* __restrict__ Z, float* __restrict__ E, int B, int N, int D,
                                                         const float* __restrict__ lengthscale) {
    __shared__ float dvec[4][32];        // per wave: the Gram's diagonal (squared row norms of the shifted rows)
    __shared__ f32x4 part[WGT ? WGT - 1 : 1][NB * (NB + 1) / 2][64];      // WGT: the partial Grams of waves 1..WGT - 1
    // (wave through readfirstlane: as a plain threadIdx.x >> 6 it is a VGPR value to the compiler, and every buffer load whose scalar offset or descriptor
    // depends on it gets a readfirstlane / compare / branch "waterfall" loop around it -- round 5, found in the ISA of the workgroup-per-task kernels)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const int b = WGT ? blockIdx.x : blockIdx.x * 4 + wave;
    if (b >= B) return;                  // (wave per task: no workgroup barrier below, the LDS scratch is per wave; WGT: b is the workgroup's)
    const brsrc zr = sm_rsrc(Z + (size_t)b * N * D, (unsigned)((size_t)N * D * 4));
    int voff[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) voff[blk] = (16 * blk + r < N) ? ((16 * blk + r) * D + 4 * g) * 4 : SM_OOB;
    const int nstep = (D + 15) >> 4;
    f32x4 x[SM_PF][NB], ref[SM_PF];
    auto load = [&](const int slot, const int s) {
        const bool in = s < nstep && 16 * s + 4 * g < D;             // D % 4 == 0: a float4 is inside the row or wholly beyond it
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) x[slot][blk] = sm_load4(zr, in ? voff[blk] : SM_OOB, s * 64);
        if (KIND != DKT_KERNEL_LINEAR) ref[slot] = sm_load4(zr, in ? 16 * g : SM_OOB, s * 64);
    };
    f32x4 acc[NB * (NB + 1) / 2];
#pragma unroll
    for (int i = 0; i < NB * (NB + 1) / 2; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float e1[XR ? XR : 1], e1x[XR ? XR : 1];                          // XR: lane (g, r): partial (row 16 + i) . (row r) and (row 16 + i) . (row 16 + r) over the features of group g
#pragma unroll
    for (int i = 0; i < (XR ? XR : 1); ++i) e1[i] = e1x[i] = 0.f;
    const int sfirst = WGT ? wave * SM_PF : 0, sstride = WGT ? WGT * SM_PF : SM_PF;
#pragma unroll
    for (int p = 0; p < SM_PF; ++p) load(p, sfirst + p);
    for (int s0 = sfirst; s0 < nstep; s0 += sstride) {
#pragma unroll
        for (int p = 0; p < SM_PF; ++p) {
            f32x4 xb[NB];
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                xb[blk] = x[p][blk];
                if (KIND != DKT_KERNEL_LINEAR) {
                    // rows beyond N loaded as 0 and must stay 0 (not -ref): the mask rides on the row test
                    const bool row_ok = 16 * blk + r < N;
#pragma unroll
                    for (int t = 0; t < 4; ++t) xb[blk][t] = row_ok ? xb[blk][t] - ref[p][t] : 0.f;
                }
            }
            load(p, s0 + p + sstride);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[0][t], xb[0][t], acc[0], 0, 0, 0);
                if constexpr (NB == 2 && XR > 0) {
#define DKT_SM_XROW(I)                                                                                  \
    if constexpr (XR > I) {                                                                             \
        const float bi = sm_bcast16<I>(xb[1][t]);                                                       \
        e1[I] = fmaf(xb[0][t], bi, e1[I]);                                                              \
        e1x[I] = fmaf(xb[1][t], bi, e1x[I]);                                                            \
    }
                    DKT_SM_XROW(0) DKT_SM_XROW(1) DKT_SM_XROW(2) DKT_SM_XROW(3)
#undef DKT_SM_XROW
                } else if constexpr (NB == 2) {
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[1][t], xb[0][t], acc[1], 0, 0, 0);      // rows 16.., columns 0..15
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[1][t], xb[1][t], acc[2], 0, 0, 0);
                }
            }
        }
    }
    if constexpr (NB == 2 && XR > 0) {   // the four feature groups of the wave meet (a butterfly: every lane ends with the same sum), then into the tile layout:
        // tile (1, 0) element [4 g + q][c] = row 16 + 4 g + q, column c: lanes g = 0, register q = e1[q]; tile (1, 1) the same with e1x (columns 16 + c)
        f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            float u = e1[i], v = e1x[i];
            u += __shfl_xor(u, 16); u += __shfl_xor(u, 32);
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            t1[i] = g == 0 ? u : 0.f;
            t2[i] = g == 0 ? v : 0.f;
        }
        acc[1] = t1;
        acc[2] = t2;
    }
    if constexpr (WGT) {                 // partial Grams of waves 1..3 -> wave 0 (fixed order: bitwise reproducible)
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < NB * (NB + 1) / 2; ++i) part[wave - 1][i][lane] = acc[i];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < WGT - 1; ++w)
#pragma unroll
            for (int i = 0; i < NB * (NB + 1) / 2; ++i) acc[i] += part[w][i][lane];
    }
    // ---- epilogue.  C / D layout: lane (g, c = r), register q  <->  element [4 g + q][c] ----
    float* Eb = E + (size_t)b * N * N;
    float inv_l2 = 0.f;
    if (KIND != DKT_KERNEL_LINEAR) {
        const float l = lengthscale[0];
        inv_l2 = 1.0f / (l * l);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            const f32x4 dg = acc[blk == 0 ? 0 : 2];
            const float dv = ((r & 3) == 0) ? dg[0] : ((r & 3) == 1) ? dg[1] : ((r & 3) == 2) ? dg[2] : dg[3];
            if ((r >> 2) == g) dvec[wave][16 * blk + r] = dv;          // element [r][r] lives in lane (g = r / 4, c = r), register r % 4
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                            // lgkmcnt(0): the wave's own LDS writes have landed
    }
    auto emit = [&](const f32x4 v, const int bi, const int bj) {      // tile (bi, bj), bi >= bj; the mirror for bi > bj
        const int col = 16 * bj + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 16 * bi + 4 * g + q;
            if (row >= N || col >= N) continue;
            float o = v[q];
            if (KIND != DKT_KERNEL_LINEAR) {
                float d2 = dvec[wave][row] + dvec[wave][col] - 2.0f * o;
                d2 = d2 > 0.f ? d2 : 0.f;
                if (row == col) d2 = 0.f;
                o = (KIND == DKT_KERNEL_RBF) ? expf(-0.5f * d2 * inv_l2) : d2 * inv_l2;
            }
            if (bi == bj) {
                if (col > row) continue;                               // lower triangle + its mirror: bitwise symmetric
                Eb[(size_t)row * N + col] = o;
                if (col != row) Eb[(size_t)col * N + row] = o;
            } else {
                Eb[(size_t)row * N + col] = o;
                Eb[(size_t)col * N + row] = o;
            }
        }
    };
    emit(acc[0], 0, 0);
    if constexpr (NB == 2) {
        emit(acc[1], 1, 0);
        emit(acc[2], 1, 1);
    }
}

// dZ[b] = s_b (W[b] + W[b]^T) Z[b], N <= 32.  WGT: a workgroup per task, wave w takes the 64-feature chunks w, w + 4, ... (as in the forward)
template <int WGT>
__global__ __launch_bounds__(WGT > 4 ? 64 * WGT : 256) void gram_small_bwd_kernel(const float* __restrict__ W, const float* __restrict__ Z, float* __restrict__ dZ,
                                                             int B, int N, int D, const float* __restrict__ ep_scale) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int b = WGT ? blockIdx.x : blockIdx.x * 4 + wave;
    if (b >= B) return;
    const float sc = ep_scale ? ep_scale[b] : 1.0f;
    const int nq = (N + 3) >> 2;                                       // k-quads: rows j = 4 kk + g of Z
    // A operands: lane (g, r = c): s (W + W^T)[16 ib + r][4 kk + g].  All 32 loads in flight at once (buffer loads, out-of-range offset outside the
    // matrix): as `(i < N && j < N) ? W[..] + W[..] : 0` the compiler branched around every pair and waited for it on the spot -- sixteen dependent memory
    // round trips at the head of every wave (round 5, found in the ISA).
    const brsrc wr = sm_rsrc(W + (size_t)b * N * N, (unsigned)((size_t)N * N * 4));
    float a[2][8];
    {
        float w1[2][8], w2[2][8];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int i = 16 * ib + c, j = 4 * kk + g;
                const bool ok = i < N && j < N;
                w1[ib][kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, ok ? (i * N + j) * 4 : SM_OOB, 0, 0));
                w2[ib][kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, ok ? (j * N + i) * 4 : SM_OOB, 0, 0));
            }
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) a[ib][kk] = sc * (w1[ib][kk] + w2[ib][kk]);
    }
    const brsrc zr = sm_rsrc(Z + (size_t)b * N * D, (unsigned)((size_t)N * D * 4));
    const brsrc dr = sm_rsrc(dZ + (size_t)b * N * D, (unsigned)((size_t)N * D * 4));
    const bool two = N > 16;
    const int nchunk = (D + 63) >> 6;
    f32x4 z0[8], z1[8];
    auto load = [&](f32x4 (&zb)[8], const int ch) {
        const bool in = ch < nchunk && 64 * ch + 4 * c < D;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)      // (unconditional: a k-quad past N loads zeros through the descriptor.  Behind the uniform `kk < nq` branch the compiler could not
            zb[kk] = sm_load4(zr, (in && 4 * kk + g < N) ? ((4 * kk + g) * D + 4 * c) * 4 : SM_OOB, ch * 256);      //  count the loads and waited `vmcnt(0)` -- for the prefetch too)
    };
    auto chunk = [&](const f32x4 (&zb)[8], const int ch) {
        f32x4 acc[2][4];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ib][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk < nq) {                                             // uniform
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][kk], zb[kk][e], acc[0][e], 0, 0, 0);
                    if (two) acc[1][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][kk], zb[kk][e], acc[1][e], 0, 0, 0);
                }
            }
        }
        // element [4 g + q][c] of tile e is dZ[16 ib + 4 g + q][64 ch + 4 c + e]: the four tiles of a lane are one float4
        const bool col_ok = 64 * ch + 4 * c < D;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
            if (ib == 1 && !two) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 16 * ib + 4 * g + q;
                const u32x4 v = {__float_as_uint(acc[ib][0][q]), __float_as_uint(acc[ib][1][q]), __float_as_uint(acc[ib][2][q]), __float_as_uint(acc[ib][3][q])};
                // (the chunk offset rides in the VGPR offset, soffset is the literal 0: see bstore4 in dkt_mfma_tiles.h)
                __builtin_amdgcn_raw_buffer_store_b128(v, dr, (row < N && col_ok) ? (row * D + 4 * c) * 4 + ch * 256 : SM_OOB, 0, 0);
            }
        }
    };
    const int c0 = WGT ? wave : 0, cs = WGT ? WGT : 1;
    load(z0, c0);
    for (int ch = c0; ch < nchunk; ch += 2 * cs) {
        load(z1, ch + cs);
        chunk(z0, ch);
        load(z0, ch + 2 * cs);
        if (ch + cs < nchunk) chunk(z1, ch + cs);
    }
}

}  // namespace

// Workgroup per task for long rows (the QMUL head's 19 x 2916), a wave per task below (a 64-feature row leaves three of the four waves idle).  Same-box
// A/B (tools/small_wg_ab.py, profiles/r05/v7_small_wg_ab.log; per 8192 tasks): backward 19 x 2916 0.941 -> 0.830 ms, 10 x 2916 0.488 -> 0.402, 25 x 1600
// 0.514 -> 0.492, 19 x 512 0.130 -> 0.133, 25 x 64 0.035 -> 0.068: from 1024 features.  Forward: equal at 8192 tasks of 19 x 2916 (0.441 / 0.438 ms),
// 0.076 -> 0.053 ms at 1024 tasks, 25 x 1600 0.261 -> 0.271: from 2048 features.  Twins library: DKT_GRAM_SMALL_WG=0 / 1 forces either form.
static int g_small_wg = -2;
void dkt_gram_small_reload_env() { g_small_wg = -2; }
static unsigned small_lds_pad() {                    // twins library: DKT_GRAM_SMALL_LDS = bytes of (unused) dynamic LDS per workgroup (the occupancy A/B; the backward's default: 56 KB)
    const char* v = dkt_variant_env("DKT_GRAM_SMALL_LDS");
    return v ? (unsigned)atoi(v) : 0u;
}
static int small_wgt(int D, int mind) {             // waves per task: 0 (a wave per task), 4, 8 (twins library, DKT_GRAM_SMALL_WG=2: measured, not the default)
    if (g_small_wg == -2) { const char* v = dkt_variant_env("DKT_GRAM_SMALL_WG"); g_small_wg = v ? atoi(v) : -1; }
    if (g_small_wg >= 0) return g_small_wg == 0 ? 0 : g_small_wg == 2 ? 8 : g_small_wg == 3 ? 16 : 4;     // (2, 3: eight / sixteen waves per task, twins library only)
    return D >= mind ? 4 : 0;
}

// Symmetric Gram of small episodes (N <= 32, D % 4 == 0, 16-byte aligned Z); returns false when it does not apply.
bool dkt_gram_small_launch(const float* Z, float* E, int B, int N, int D, int kind, const float* lengthscale, hipStream_t st) {
    if (N > 32 || (D & 3) || ((uintptr_t)Z & 15) || (size_t)N * D * 4 >= 0x7fffff00ull) return false;
    // workgroup per task: the forward from 2048 features (16 < N <= 32, linear / RBF: the QMUL head), see small_wgt()
    const int wgt = (N > 16 && kind != DKT_KERNEL_SQDIST) ? small_wgt(D, 2048) : 0;
    const dim3 grid(wgt ? B : (B + 3) / 4), block(wgt > 4 ? 64 * wgt : 256);
    unsigned pad = small_lds_pad();
    // rows beyond sixteen on the VALU: the default for RBF (tools/small_xr_ab.py, profiles/r06/small_xr_ab.log; per 8192 tasks of 19 x 2916 0.396 -> 0.394 ms, 20 x 2916
    // 0.398 -> 0.372, 19 x 512 0.074 -> 0.071, 1024 tasks of 19 x 2916 0.049 -> 0.040); the linear kernel (no row shift on the VALU) measured 0.369 -> 0.387 and keeps
    // its MFMA tiles.  Twins library: DKT_GRAM_SMALL_XR = 0 / 1 = neither / both kinds.
    bool xr_on = kind == DKT_KERNEL_RBF;
    { const char* v = dkt_variant_env("DKT_GRAM_SMALL_XR"); if (v) xr_on = atoi(v) != 0; }
    const int xr = (xr_on && N > 16 && N <= 20 && (wgt == 0 || wgt == 4) && kind != DKT_KERNEL_SQDIST) ? (N <= 19 ? 3 : 4) : 0;
    // the XR instance of the four-wave kernel (91 registers: five workgroups per CU) streams best at THREE per CU once every CU has a queue of tasks (tools/small_pf_ab.py,
    // profiles/r06/small_pf_ab.log: 8192 tasks of 19 x 2916 0.381 -> 0.357 ms, 0.386 at two; 1024 tasks: 0.039 -> 0.043, no cap there): 40 KB of untouched dynamic LDS
    if (xr && wgt == 4 && B >= 4096 && !dkt_variant_env("DKT_GRAM_SMALL_LDS")) pad = 40960u;
#define DKT_SM_LAUNCH(K)                                                                                              \
    do {                                                                                                              \
        if (N <= 16) hipLaunchKernelGGL((gram_small_kernel<K, 1, 0>), grid, block, pad, st, Z, E, B, N, D, lengthscale);   \
        else hipLaunchKernelGGL((gram_small_kernel<K, 2, 0>), grid, block, pad, st, Z, E, B, N, D, lengthscale);           \
    } while (0)
#ifdef DKT_TWINS
    if (wgt == 8 && kind == DKT_KERNEL_LINEAR) hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_LINEAR, 2, 8>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
    else if (wgt == 8 && kind == DKT_KERNEL_RBF) hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_RBF, 2, 8>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
    else if (wgt == 16 && kind == DKT_KERNEL_LINEAR) hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_LINEAR, 2, 16>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
    else if (wgt == 16 && kind == DKT_KERNEL_RBF) hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_RBF, 2, 16>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
    else
#endif
    // 17 .. 20 rows (the QMUL head: 19): the rows beyond sixteen on the VALU (template parameter XR), see the kernel.  Twins library: DKT_GRAM_SMALL_XR=0 = MFMA tiles
#ifdef DKT_TWINS
    if (xr && wgt && kind == DKT_KERNEL_LINEAR) {
        if (xr == 3) hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_LINEAR, 2, 4, 3>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
        else hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_LINEAR, 2, 4, 4>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
    } else if (xr && kind == DKT_KERNEL_LINEAR) {
        if (xr == 3) hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_LINEAR, 2, 0, 3>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
        else hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_LINEAR, 2, 0, 4>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
    } else
#endif
    if (xr && wgt && kind == DKT_KERNEL_RBF) {
        if (xr == 3) hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_RBF, 2, 4, 3>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
        else hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_RBF, 2, 4, 4>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
    } else if (xr && kind == DKT_KERNEL_RBF) {
        if (xr == 3) hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_RBF, 2, 0, 3>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
        else hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_RBF, 2, 0, 4>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
    } else
    if (wgt && kind == DKT_KERNEL_LINEAR) hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_LINEAR, 2, 4>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
    else if (wgt && kind == DKT_KERNEL_RBF) hipLaunchKernelGGL((gram_small_kernel<DKT_KERNEL_RBF, 2, 4>), grid, block, pad, st, Z, E, B, N, D, lengthscale);
    else if (kind == DKT_KERNEL_LINEAR) DKT_SM_LAUNCH(DKT_KERNEL_LINEAR);
    else if (kind == DKT_KERNEL_RBF) DKT_SM_LAUNCH(DKT_KERNEL_RBF);
    else if (kind == DKT_KERNEL_SQDIST) DKT_SM_LAUNCH(DKT_KERNEL_SQDIST);
    else return false;
#undef DKT_SM_LAUNCH
    return true;
}

bool dkt_gram_small_bwd_launch(const float* W, const float* Z, float* dZ, int B, int N, int D, const float* sc, hipStream_t st) {
    if (N > 32 || (D & 3) || ((uintptr_t)Z & 15) || ((uintptr_t)dZ & 15) || (size_t)N * D * 4 >= 0x7fffff00ull) return false;
    const int wgt = small_wgt(D, 1024);
    // Workgroups per CU (round 6, tools/small_occ_ab.py, profiles/r06/small_occ_ab.log): the four-wave workgroup-per-task kernel streams FASTER at one or two workgroups
    // per CU than at the three its registers allow -- 8192 tasks of 19 x 2916: 0.880 -> 0.801 ms, 10 x 2916: 0.400 -> 0.380, 32 x 2048: 0.846 -> 0.830, 25 x 1600:
    // 0.499 -> 0.491 (two chunks of 19 rows x 256 B in flight per wave already cover the memory latency at eight waves per CU; more tasks in flight only add streams
    // for the DRAM pages) -- so the launch reserves 56 KB of dynamic LDS it never touches: at most two workgroups per CU.  Bitwise the same results.
    unsigned pad = wgt == 4 ? 57344u : 0u;
#ifdef DKT_TWINS
    if (dkt_variant_env("DKT_GRAM_SMALL_LDS")) pad = small_lds_pad();
    if (wgt == 8) hipLaunchKernelGGL(gram_small_bwd_kernel<8>, dim3(B), dim3(512), pad, st, W, Z, dZ, B, N, D, sc);
    else if (wgt == 16) hipLaunchKernelGGL(gram_small_bwd_kernel<16>, dim3(B), dim3(1024), pad, st, W, Z, dZ, B, N, D, sc);
    else
#endif
    if (wgt) hipLaunchKernelGGL(gram_small_bwd_kernel<4>, dim3(B), dim3(256), pad, st, W, Z, dZ, B, N, D, sc);
    else hipLaunchKernelGGL(gram_small_bwd_kernel<0>, dim3((B + 3) / 4), dim3(256), pad, st, W, Z, dZ, B, N, D, sc);
    return true;
}
