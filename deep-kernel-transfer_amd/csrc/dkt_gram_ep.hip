// dkt_gram_ep.hip -- EPISODE-RESIDENT Gram kernels for the few-shot shapes (64 < N <= 128, D % 4 == 0).
//
// gram_sym_ep_kernel<NT, BK>: one 256-thread workgroup per episode streams Z[N, D] through LDS exactly
//   once ([16 NT rows] x BK floats per stage, 16-byte coalesced loads, register prefetch + double-buffered
//   LDS, one barrier per stage) and computes ONLY the NT(NT+1)/2 lower 16x16 tiles of Z Z^T with
//   v_mfma_f32_16x16x4_f32 -- 28 tiles for N = 105 instead of the 48 tile-equivalents of a 64x64 tiling.
//   The tiles are split statically over the 4 waves by tile ROW (dkt_tiles.h), so a wave reads at most
//   NT b128 fragments per 16-wide K slice for its ~NT+1 tiles x 4 MFMAs and no cross-wave reduction is
//   needed.  HBM traffic = the algorithmic 4 (N D + N^2) bytes per episode.
//
// gram_bwd_ep_kernel<NT, BD>: dZ = s_b (W + W^T) Z, one workgroup of NT waves per episode.  Wave w keeps the
//   A-operand fragments of row block w of (W + W^T) in registers for the whole episode (NT K-slices x 4
//   VGPRs), Z is streamed through LDS in [16 NT] x BD column slabs (read once, coalesced), every wave
//   multiplies its row block with the shared slab.  HBM traffic = 4 (2 N D + N^2) bytes per episode.
//
// Replaces (same lines as dkt_gram.hip): ExactGPLayer.forward -> covar_module(x) (methods/DKT.py:375-378)
// and autograd through it (DKT.py:163).
#include "dkt_common.h"
#include "dkt_tiles.h"
#include "dkt_split.h"
#include "../../include/dkt_abi.h"

namespace {

// ---------------------------------------------------------------------------------------------
template <int NT, int RA, int RB, int BK>
__device__ __forceinline__ void sym_tiles_mfma(f32x4* acc, const float* zs, int r16, int q) {
    constexpr int LD = BK + 8;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
        f32x4 fr[RA + 1];
#pragma unroll
        for (int blk = 0; blk <= RA; ++blk)
            fr[blk] = *reinterpret_cast<const f32x4*>(&zs[(blk * 16 + r16) * LD + kk * 16 + 4 * q]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int tj = 0; tj <= RA; ++tj)
                acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[RA][t], fr[tj][t], acc[tj], 0, 0, 0);
            if constexpr (RB >= 0) {
#pragma unroll
                for (int tj = 0; tj <= RB; ++tj)
                    acc[RA + 1 + tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[RB][t], fr[tj][t], acc[RA + 1 + tj], 0, 0, 0);
            }
        }
    }
}

template <int NT, int BK>
__global__ __launch_bounds__(256) void gram_sym_ep_kernel(const float* __restrict__ Z, float* __restrict__ E, int N, int D) {
    constexpr int NP = 16 * NT;
    constexpr int LD = BK + 8;                         // (LD/4) mod 16 in {2, 10}: conflict-free b128 fragment reads
    constexpr int V4_PER_ROW = BK / 4;
    constexpr int NV4 = NP * V4_PER_ROW;               // float4 per stage
    constexpr int NLD = (NV4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float zs[2][NP * LD];

    const int b = blockIdx.x;
    const float* Zb = Z + (size_t)b * N * D;
    float* Eb = E + (size_t)b * N * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;

    float4 rg[NLD];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / V4_PER_ROW, c4 = idx % V4_PER_ROW;
            const int k = k0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((NV4 % 256 == 0 || idx < NV4) && row < N && k < D) v = *reinterpret_cast<const float4*>(Zb + (size_t)row * D + k);
            rg[i] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / V4_PER_ROW, c4 = idx % V4_PER_ROW;
            if (NV4 % 256 == 0 || idx < NV4) *reinterpret_cast<float4*>(&zs[buf][row * LD + 4 * c4]) = rg[i];
        }
    };

    f32x4 acc[NT + 1];
#pragma unroll
    for (int i = 0; i <= NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (D + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        if (wave == 0) {
            if constexpr (RowsOf<NT, 0>::RA >= 0) sym_tiles_mfma<NT, RowsOf<NT, 0>::RA, RowsOf<NT, 0>::RB, BK>(acc, zs[buf], r16, q);
        } else if (wave == 1) {
            if constexpr (RowsOf<NT, 1>::RA >= 0) sym_tiles_mfma<NT, RowsOf<NT, 1>::RA, RowsOf<NT, 1>::RB, BK>(acc, zs[buf], r16, q);
        } else if (wave == 2) {
            if constexpr (RowsOf<NT, 2>::RA >= 0) sym_tiles_mfma<NT, RowsOf<NT, 2>::RA, RowsOf<NT, 2>::RB, BK>(acc, zs[buf], r16, q);
        } else {
            if constexpr (RowsOf<NT, 3>::RA >= 0) sym_tiles_mfma<NT, RowsOf<NT, 3>::RA, RowsOf<NT, 3>::RB, BK>(acc, zs[buf], r16, q);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    if (wave == 0) {
        if constexpr (RowsOf<NT, 0>::RA >= 0) sym_store_row<RowsOf<NT, 0>::RA>(acc, Eb, N, r16, q);
        if constexpr (RowsOf<NT, 0>::RB >= 0) sym_store_row<RowsOf<NT, 0>::RB>(acc + RowsOf<NT, 0>::RA + 1, Eb, N, r16, q);
    } else if (wave == 1) {
        if constexpr (RowsOf<NT, 1>::RA >= 0) sym_store_row<RowsOf<NT, 1>::RA>(acc, Eb, N, r16, q);
        if constexpr (RowsOf<NT, 1>::RB >= 0) sym_store_row<RowsOf<NT, 1>::RB>(acc + RowsOf<NT, 1>::RA + 1, Eb, N, r16, q);
    } else if (wave == 2) {
        if constexpr (RowsOf<NT, 2>::RA >= 0) sym_store_row<RowsOf<NT, 2>::RA>(acc, Eb, N, r16, q);
        if constexpr (RowsOf<NT, 2>::RB >= 0) sym_store_row<RowsOf<NT, 2>::RB>(acc + RowsOf<NT, 2>::RA + 1, Eb, N, r16, q);
    } else {
        if constexpr (RowsOf<NT, 3>::RA >= 0) sym_store_row<RowsOf<NT, 3>::RA>(acc, Eb, N, r16, q);
        if constexpr (RowsOf<NT, 3>::RB >= 0) sym_store_row<RowsOf<NT, 3>::RB>(acc + RowsOf<NT, 3>::RA + 1, Eb, N, r16, q);
    }
}

// ---------------------------------------------------------------------------------------------
// gram_sym_ep_split_kernel<NT>: the same episode-resident symmetric Gram on the bf16 MFMA pipe.
// Every fp32 feature is split EXACTLY into three bf16 pieces x = h + m + l (8 + 8 + 8 significand bits) while it is
// staged into LDS (v_cvt_pk_bf16_f32 + subtract, three bf16 planes), and each fp32 product is rebuilt from the six
// leading cross terms  hh + hm + mh + hl + lh + mm  with v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  The dropped
// terms (ml, lm, ll) are <= 2^-23 relative per product -- the size of one fp32 rounding -- so the result is
// fp32-faithful, while the MFMA time per 32-wide K slice drops from 8 x 32 to 6 x ~17 cycles per tile: the kernel
// leaves the fp32-MFMA roof (157 TF) and becomes HBM-bound.
// NBUF = LDS stage buffers (2: one barrier per stage, 2 workgroups/CU; 1: two barriers, 4 workgroups/CU);
// PF   = global-load run-ahead in stages (register sets): PF stages x 14 KB per workgroup stay in flight.
// SPL = 3: the bf16 split above (any operand range).  SPL = 2: 2-way f16 split of features scaled by 2^15 (dkt_split.h), for
// rows the caller declares bounded by 1 in magnitude (DKT_KERNEL_LINEAR_UNIT: the cossim / bncossim features after
// F.normalize); an element beyond 1.999 overflows f16 and poisons the episode with inf / NaN (loud, not silent).
// The MFMA flushes f16 subnormals: the low piece of an element below 2^-18 (and all of one below 2^-29) is dropped, an
// absolute error of at most 2^-30 |b| per product -- far below the fp32 resolution of a cosine similarity.
#define DKT_F16_UNSCALE (1.f / (32768.f * 32768.f))
template <int NT, int NBUF, int PF, int BK = 32, int SPL = 3, int MINWG = ((NBUF == 1 && NT <= 7 && BK == 32) ? (PF == 1 ? 4 : 3) : 2), int POL = 0>
__global__ __launch_bounds__(256, MINWG) void gram_sym_ep_split_kernel(const float* __restrict__ Z, float* __restrict__ E, int N, int D) {
    constexpr int NP = 16 * NT;
    constexpr int SPLD = BK + 16;
    constexpr int V4_PER_ROW = BK / 4;
    constexpr int NV4 = NP * V4_PER_ROW;
    constexpr int NLD = (NV4 + 255) / 256;
    // the LDS image has NLD * 256 / V4_PER_ROW >= NP rows, so that EVERY thread stages exactly NLD float4 with no
    // exec-masked tail: straight-line staging code keeps the compiler's vmcnt bookkeeping exact, which is what lets the
    // far prefetch (PF = 2) really stay in flight across the MFMA phase.  Rows >= N load as zeros (out-of-range offset).
    constexpr int NPL = NLD * 256 / V4_PER_ROW;
    constexpr int PLANE = NPL * SPLD;
    __shared__ __attribute__((aligned(16))) __bf16 zp[NBUF][SPL * PLANE];

    const int b = blockIdx.x;
    const float* Zb = Z + (size_t)b * N * D;
    float* Eb = E + (size_t)b * N * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;

    // Buffer loads: one 32-bit per-lane byte offset per staged float4 (rows >= N get an out-of-range offset and read
    // as 0 through the descriptor's bounds check), the slice offset k0 rides in the SGPR operand -- no 64-bit address
    // arithmetic and no exec-mask juggling in the steady state; only a ragged last slice (D % 32 != 0) tests k < D.
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Zb), 0, N * D * 4, 0x00020000);
    int voff[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / V4_PER_ROW, c4 = idx % V4_PER_ROW;
        voff[i] = (row < N) ? (row * D + 4 * c4) * 4 : 0x7ffffff0;
    }
    auto gload = [&](float4 (&rg)[NLD], int k0) {
        const bool ragged = k0 + BK > D;                 // uniform
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int vo = voff[i];
            if (ragged) vo = (k0 + 4 * ((tid + 256 * i) % V4_PER_ROW) < D) ? vo : 0x7ffffff0;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(zr, vo, k0 * 4, (POL & 2) ? 2 : 0);
            rg[i] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto lstore = [&](const float4 (&rg)[NLD], int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / V4_PER_ROW, c4 = idx % V4_PER_ROW;
            if constexpr (SPL == 3) {
                bf16x4 h, m, l;
                split3(rg[i], h, m, l);
                __bf16* dst = &zp[buf][row * SPLD + 4 * c4];
                *reinterpret_cast<bf16x4*>(dst) = h;
                *reinterpret_cast<bf16x4*>(dst + PLANE) = m;
                *reinterpret_cast<bf16x4*>(dst + 2 * PLANE) = l;
            } else {
                f16x4 h, m;
                split2h(rg[i], DKT_F16_SCALE, h, m);
                _Float16* dst = reinterpret_cast<_Float16*>(&zp[buf][row * SPLD + 4 * c4]);
                *reinterpret_cast<f16x4*>(dst) = h;
                *reinterpret_cast<f16x4*>(dst + PLANE) = m;
            }
        }
    };

    f32x4 acc[NT + 1];
#pragma unroll
    for (int i = 0; i <= NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto tiles = [&](auto rows, const __bf16* zs) {
        using R = decltype(rows);
        if constexpr (R::RA >= 0) {
            if constexpr (SPL == 3) sym_tiles_mfma_bf16x3<NT, R::RA, R::RB, SPLD, PLANE>(acc, zs, r16, q);
            else sym_tiles_mfma_f16x2<NT, R::RA, R::RB, SPLD, PLANE>(acc, reinterpret_cast<const _Float16*>(zs), r16, q);
        }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            const __bf16* zs = zp[buf] + 32 * ks;
            if (wave == 0) tiles(RowsOf<NT, 0>{}, zs);
            else if (wave == 1) tiles(RowsOf<NT, 1>{}, zs);
            else if (wave == 2) tiles(RowsOf<NT, 2>{}, zs);
            else tiles(RowsOf<NT, 3>{}, zs);
        }
    };

    const int nk = (D + BK - 1) / BK;
    float4 r0[NLD], r1[NLD];
#if defined(DKT_EXP_CLOCKS)
    long long clk[6] = {0, 0, 0, 0, 0, 0};
    const long long cstart = clock64();
#endif
    // stage kt: LDS holds slice kt, `rnear` holds slice kt+1 (PF = 2) or is loaded now (PF = 1), `rfar` is issued for kt+2.
    // PF = 2 runs with NO conditionals around loads and stores (slices past D load as zeros through the ragged test and
    // are staged and multiplied as zeros; an odd slice count is rounded up): on a branch-free path the compiler's vmcnt
    // values are exact, so the 4 far loads really stay in flight while the near ones are consumed.
    auto stage = [&](float4 (&rnear)[NLD], float4 (&rfar)[NLD], int kt) {
        const int buf = (NBUF == 2) ? (kt & 1) : 0;
#if defined(DKT_EXP_CLOCKS)
        // phase clocks of wave 0 (measurement build only): issue loads | MFMA phase | barrier 1 | load wait | split+store | barrier 2
        const long long c0 = clock64();
#endif
        if constexpr (PF == 2) {
            gload(rfar, (kt + 2) * BK);
        } else {
            if (kt + 1 < nk) gload(rnear, (kt + 1) * BK);
        }
#if defined(DKT_EXP_CLOCKS)
        const long long c1 = clock64();
#endif
        compute(buf);
#if defined(DKT_EXP_CLOCKS)
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): fragment reads done (MFMAs may still execute)
        const long long c2 = clock64();
#endif
        if constexpr (NBUF == 1) __syncthreads();
#if defined(DKT_EXP_CLOCKS)
        const long long c3 = clock64();
        __builtin_amdgcn_s_waitcnt(0x0070 | 0x0f00);        // vmcnt(0)
        const long long c4 = clock64();
#endif
        if constexpr (PF == 2) {
            lstore(rnear, (NBUF == 2) ? (buf ^ 1) : 0);
        } else {
            if (kt + 1 < nk) lstore(rnear, (NBUF == 2) ? (buf ^ 1) : 0);
        }
#if defined(DKT_EXP_CLOCKS)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const long long c5 = clock64();
#endif
        __syncthreads();
#if defined(DKT_EXP_CLOCKS)
        const long long c6 = clock64();
        clk[0] += c1 - c0; clk[1] += c2 - c1; clk[2] += c3 - c2; clk[3] += c4 - c3; clk[4] += c5 - c4; clk[5] += c6 - c5;
#endif
    };
    gload(r0, 0);
    if constexpr (PF == 2) gload(r1, BK);
    lstore(r0, 0);
    __syncthreads();
    if constexpr (PF == 2) {
        for (int kt = 0; kt < nk; kt += 2) {
            stage(r1, r0, kt);
            stage(r0, r1, kt + 1);
        }
    } else {
        for (int kt = 0; kt < nk; ++kt) stage(r0, r1, kt);
    }
#if defined(DKT_EXP_CLOCKS)
    const long long cloop = clock64();
#endif
    constexpr float OUT_SCALE = (SPL == 3) ? 1.f : DKT_F16_UNSCALE;
    auto store = [&](auto rows) {
        using R = decltype(rows);
        if constexpr (R::RA >= 0) sym_store_row<R::RA>(acc, Eb, N, r16, q, OUT_SCALE);
        if constexpr (R::RB >= 0) sym_store_row<R::RB>(acc + R::RA + 1, Eb, N, r16, q, OUT_SCALE);
    };
    if (wave == 0) store(RowsOf<NT, 0>{});
    else if (wave == 1) store(RowsOf<NT, 1>{});
    else if (wave == 2) store(RowsOf<NT, 2>{});
    else store(RowsOf<NT, 3>{});
#if defined(DKT_EXP_CLOCKS)
    if (tid == 0) {
        // the clock dump overwrites the (unused in this build) first row of E[b]: E[b][0..7] as integer kilo-ticks
        const long long cend = clock64();
        float* dbg = Eb;
        for (int i = 0; i < 6; ++i) dbg[i] = (float)clk[i];
        dbg[6] = (float)(cloop - cstart);
        dbg[7] = (float)(cend - cloop);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// dZ = s (W + W^T) Z.  The workgroup has NT waves; wave w owns output row block w (16 rows) for every
// column of the slab, keeps its NT A-fragments of s (W + W^T) in registers for the whole episode, and all
// waves share the Z slab in LDS -- perfectly balanced MFMA work, BD/16 float4 staging loads per thread.
template <int NT, int BD>
__global__ __launch_bounds__(64 * NT) void gram_bwd_ep_kernel(const float* __restrict__ W, const float* __restrict__ Z,
                                                              float* __restrict__ dZ, int N, int D,
                                                              const float* __restrict__ ep_scale) {
    constexpr int NP = 16 * NT;
    constexpr int NTH = 64 * NT;
    constexpr int BLD = BD + 4;                        // b32 B-fragment reads: rows 4q+t, 16 consecutive columns
    constexpr int V4_PER_ROW = BD / 4;
    constexpr int NCT = BD / 16;                       // column tiles per slab == float4 loads per thread per slab
    __shared__ __attribute__((aligned(16))) float zs[2][NP * BLD];

    const int b = blockIdx.x;
    const float* Wb = W + (size_t)b * N * N;
    const float* Zb = Z + (size_t)b * N * D;
    float* dZb = dZ + (size_t)b * N * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const float s = ep_scale ? ep_scale[b] : 1.0f;

    // A fragments: a[kk][t] = s * Wsym[wave*16 + r16][16 kk + 4 q + t]
    f32x4 afr[NT];
    {
        const int row = wave * 16 + r16;
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k = kk * 16 + 4 * q + t;
                float v = 0.f;
                if (row < N && k < N) v = s * (Wb[row * N + k] + Wb[k * N + row]);
                afr[kk][t] = v;
            }
        }
    }

    float4 rg[NCT];
    auto gload = [&](int d0) {
#pragma unroll
        for (int i = 0; i < NCT; ++i) {
            const int idx = tid + NTH * i;
            const int row = idx / V4_PER_ROW, c4 = idx % V4_PER_ROW;
            const int d = d0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < N && d < D) v = *reinterpret_cast<const float4*>(Zb + (size_t)row * D + d);
            rg[i] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCT; ++i) {
            const int idx = tid + NTH * i;
            const int row = idx / V4_PER_ROW, c4 = idx % V4_PER_ROW;
            *reinterpret_cast<float4*>(&zs[buf][row * BLD + 4 * c4]) = rg[i];
        }
    };

    const int nslab = (D + BD - 1) / BD;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int sl = 0; sl < nslab; ++sl) {
        const int buf = sl & 1;
        if (sl + 1 < nslab) gload((sl + 1) * BD);
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* bs = zs[buf];
#pragma unroll
        for (int kk = 0; kk < NT; ++kk) {
            f32x4 bf[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int t = 0; t < 4; ++t) bf[ct][t] = bs[(kk * 16 + 4 * q + t) * BLD + ct * 16 + r16];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[kk][t], bf[ct][t], acc[ct], 0, 0, 0);
        }
        const int d0 = sl * BD;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = wave * 16 + 4 * q + reg;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int d = d0 + ct * 16 + r16;
                if (row < N && d < D) dZb[(size_t)row * D + d] = acc[ct][reg];
            }
        }
        if (sl + 1 < nslab) lstore(buf ^ 1);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// gram_bwd_ep_bf16x3_kernel<NT>: dZ = s (W + W^T) Z on the bf16 MFMA pipe with the same exact 3-way split.
// The contraction index j is the SLOW index of Z in memory, so the Z slab is transposed while it is staged:
// a thread loads a 4(j) x 4(d) fp32 block (4 float4, 256-B coalesced rows), splits it, and writes, per d and per
// plane, the 4 consecutive-j bf16 as one 8-byte LDS store into a [d][j] image -- one ds_read_b128 then yields a
// lane's 8 consecutive-j B-operand values.  LDS row of feature d = 4 r + t is 16 t + r: tile t of the slab holds
// the 16 features {4 r + t}, which makes (i) the 8-byte stores of a wave cover 2 full bank rows (the minimum),
// (ii) the fragment reads conflict-free (row stride == 2 mod 4 sixteen-byte units) and (iii) the epilogue a
// float4 store per accumulator register (tiles t = 0..3 of one lane are 4 consecutive d).
// Wave w owns output rows [16 w, 16 w + 16) and keeps its A fragments (3 planes of s (W + W^T), K padded to a
// multiple of 32) in registers for the whole episode.
template <int NT, int NBUF, int PF>
__global__ __launch_bounds__(64 * NT, NBUF == 1 ? 4 : 2) void gram_bwd_ep_bf16x3_kernel(const float* __restrict__ W, const float* __restrict__ Z,
                                                                     float* __restrict__ dZ, int N, int D,
                                                                     const float* __restrict__ ep_scale) {
    constexpr int NP = 16 * NT;
    constexpr int NTH = 64 * NT;
    constexpr int BD = 64;                               // features per slab: one 4x4 staging block per thread
    constexpr int KS = (NP + 31) / 32;                   // k32 slices
    constexpr int KP = 32 * KS;
    constexpr int SU = (KP / 8) + ((KP / 8) % 4 == 2 ? 0 : (6 - (KP / 8) % 4) % 4);   // 16-B units per LDS row, == 2 mod 4
    constexpr int RS = 8 * SU;                           // bf16 per LDS row
    constexpr int PLANE = BD * RS;
    static_assert(SU % 4 == 2 && RS >= KP, "LDS row stride");
    __shared__ __attribute__((aligned(16))) __bf16 zt[NBUF][3 * PLANE];

    const int b = blockIdx.x;
    const float* Wb = W + (size_t)b * N * N;
    const float* Zb = Z + (size_t)b * N * D;
    float* dZb = dZ + (size_t)b * N * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const float s = ep_scale ? ep_scale[b] : 1.0f;

    const int d4 = tid & 15, jg = tid >> 4;              // staging block: rows 4 jg .. 4 jg + 3, features 4 d4 .. 4 d4 + 3
    // buffer loads: rows j >= N and features d >= D get an out-of-range offset and read as zeros (no exec-masked code)
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Zb), 0, N * D * 4, 0x00020000);
    int voff[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) voff[rr] = (4 * jg + rr < N) ? ((4 * jg + rr) * D + 4 * d4) * 4 : 0x7ffffff0;
    auto gload = [&](float4 (&rg)[4], int d0) {
        const bool in = d0 + 4 * d4 < D;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(zr, in ? voff[rr] : 0x7ffffff0, d0 * 4, 0);
            rg[rr] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    const int nslab = (D + BD - 1) / BD;
    float4 r0[4], r1[4];
    gload(r0, 0);                                        // the first slab(s) fly while W is staged and split
    if constexpr (PF == 2) gload(r1, BD);

    // A fragments: slot e of slice ks on lane (r16, q) is k = 32 ks + 8 q + e.  W[b] (N x N fp32, 44 KB) is first
    // copied into the (still unused) staging LDS with coalesced loads, so the row AND the column access of
    // s (W + W^T) are LDS reads instead of 2 x 8 KS scattered global loads per lane.
    bf16x8 ah[KS], am[KS], al[KS];
    {
        float* wl = reinterpret_cast<float*>(&zt[0][0]);
        static_assert(sizeof(zt) >= NP * NP * 4, "W does not fit the staging buffer(s)");
        const int nn = N * N;
        DKT_LDS_STAGE_OLD_LOOP(for (int i = tid; i < nn; i += NTH) wl[i] = Wb[i];)
        {
            LdsStage<NTH, NT> wst;                      // all of W in flight at once (dkt_split.h)
            wst.load(Wb, nn, tid);
            wst.store(wl, nn, tid);
        }
        __syncthreads();
        const int row = wave * 16 + r16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 32 * ks + 8 * q + e;
                float v = 0.f;
                if (row < N && k < N) v = s * (wl[row * N + k] + wl[k * N + row]);
                __bf16 h, m, l;
                split3s(v, h, m, l);
                ah[ks][e] = h;
                am[ks][e] = m;
                al[ks][e] = l;
            }
        }
        __syncthreads();
    }
    // columns j in [NP, KP) of the [d][j] image are never staged: zero them once (NaN bit patterns would poison 0 * x)
    if constexpr (KP > NP) {
        constexpr int PADV = (KP - NP) / 8;              // 16-byte pieces per row
        for (int i = tid; i < NBUF * 3 * BD * PADV; i += NTH) {
            const int rowi = i / PADV, pc = i % PADV;    // rowi enumerates (buffer, plane, d)
            __bf16* dst = &zt[0][0] + (size_t)rowi * RS + NP + 8 * pc;
            *reinterpret_cast<float4*>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    auto lstore = [&](const float4 (&rg)[4], int buf) {
        const float x[4][4] = {{rg[0].x, rg[1].x, rg[2].x, rg[3].x}, {rg[0].y, rg[1].y, rg[2].y, rg[3].y},
                               {rg[0].z, rg[1].z, rg[2].z, rg[3].z}, {rg[0].w, rg[1].w, rg[2].w, rg[3].w}};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bf16x4 h, m, l;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                __bf16 hh, mm, ll;
                split3s(x[t][rr], hh, mm, ll);
                h[rr] = hh;
                m[rr] = mm;
                l[rr] = ll;
            }
            __bf16* dst = &zt[buf][(16 * t + d4) * RS + 4 * jg];
            *reinterpret_cast<bf16x4*>(dst) = h;
            *reinterpret_cast<bf16x4*>(dst + PLANE) = m;
            *reinterpret_cast<bf16x4*>(dst + 2 * PLANE) = l;
        }
    };
    auto compute_store = [&](int buf, int d0) {
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const __bf16* base = &zt[buf][r16 * RS + 8 * q];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const __bf16* p = base + 16 * t * RS + 32 * ks;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(p);
                const bf16x8 bm = *reinterpret_cast<const bf16x8*>(p + PLANE);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(p + 2 * PLANE);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[ks], bm, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bm, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[ks], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bh, acc[t], 0, 0, 0);
            }
        }
        const int d = d0 + 4 * r16;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = wave * 16 + 4 * q + reg;
            if (row < N && d < D)
                *reinterpret_cast<float4*>(dZb + (size_t)row * D + d) = make_float4(acc[0][reg], acc[1][reg], acc[2][reg], acc[3][reg]);
        }
    };

    // slab sl: LDS holds slab sl, `rnear` holds slab sl+1 (PF = 2) or is loaded now (PF = 1), `rfar` is issued for sl+2
    auto stage = [&](float4 (&rnear)[4], float4 (&rfar)[4], int sl) {
        const int buf = (NBUF == 2) ? (sl & 1) : 0;
        if constexpr (PF == 2) {
            gload(rfar, (sl + 2) * BD);                  // unconditional: slabs past D load (and stage, and multiply) as zeros
        } else {
            if (sl + 1 < nslab) gload(rnear, (sl + 1) * BD);
        }
        compute_store(buf, sl * BD);
        if constexpr (NBUF == 1) __syncthreads();
        if constexpr (PF == 2) {
            lstore(rnear, (NBUF == 2) ? (buf ^ 1) : 0);
        } else {
            if (sl + 1 < nslab) lstore(rnear, (NBUF == 2) ? (buf ^ 1) : 0);
        }
        __syncthreads();
    };
    __syncthreads();                                     // pad columns zeroed
    lstore(r0, 0);
    __syncthreads();
    if constexpr (PF == 2) {
        for (int sl = 0; sl < nslab; sl += 2) {
            stage(r1, r0, sl);
            stage(r0, r1, sl + 1);                       // an odd slab count runs one all-zero slab (its stores are masked)
        }
    } else {
        for (int sl = 0; sl < nslab; ++sl) stage(r0, r1, sl);
    }
}

// ---------------------------------------------------------------------------------------------
// gram_bwd_ep_f16x2_kernel<NT>: dZ = s (W + W^T) Z with the 2-way scaled-f16 split (dkt_split.h) for unit-norm rows of Z
// (DKT_GRAM_UNIT_ROWS).  Same data movement as the bf16 kernel above; differences:
//   * Z is scaled by 2^15 and split into two f16 planes (two thirds of the LDS stores, half the MFMAs);
//   * the A operand s (W + W^T) has no a-priori range, so every ROW is scaled by its own power of two (row maximum ->
//     [2^14, 2^15)) before the split -- exact, and undone per output row in the epilogue together with the 2^-15 of Z;
//   * the staging copy of W shares the LDS with both stage buffers (74 KB, 2 workgroups per CU).
//   * POL (measurement variants): bit 0 = non-temporal dZ stores, bit 1 = non-temporal Z loads (aux = 2)
// (waves per SIMD the register allocation must leave room for: two workgroups of NT waves per CU -- 4 at NT >= 7, i.e. <= 128 VGPRs)
template <int NT, int NBUF, int PF, int POL = 0>
__global__ __launch_bounds__(64 * NT, (2 * NT + 3) / 4) void gram_bwd_ep_f16x2_kernel(const float* __restrict__ W, const float* __restrict__ Z,
                                                                      float* __restrict__ dZ, int N, int D,
                                                                      const float* __restrict__ ep_scale) {
    constexpr int NP = 16 * NT;
    constexpr int NTH = 64 * NT;
    constexpr int BD = 64;
    constexpr int KS = (NP + 31) / 32;
    constexpr int KP = 32 * KS;
    constexpr int SU = (KP / 8) + ((KP / 8) % 4 == 2 ? 0 : (6 - (KP / 8) % 4) % 4);
    constexpr int RS = 8 * SU;
    constexpr int PLANE = BD * RS;
    static_assert(SU % 4 == 2 && RS >= KP, "LDS row stride");
    constexpr int STAGE_BYTES = NBUF * 2 * PLANE * 2;
    constexpr int LDS_BYTES = STAGE_BYTES > NP * NP * 4 ? STAGE_BYTES : NP * NP * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    __shared__ float rowinv[NP];
    _Float16* zt = reinterpret_cast<_Float16*>(lds);     // [NBUF][2 * PLANE]

    const int b = blockIdx.x;
    const float* Wb = W + (size_t)b * N * N;
    const float* Zb = Z + (size_t)b * N * D;
    float* dZb = dZ + (size_t)b * N * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const float s = ep_scale ? ep_scale[b] : 1.0f;

    const int d4 = tid & 15, jg = tid >> 4;
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Zb), 0, N * D * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t dzr = __builtin_amdgcn_make_buffer_rsrc(dZb, 0, N * D * 4, 0x00020000);
    // one offset register for the thread's four rows: a row past N starts behind the descriptor's end by itself (the range check sees voffset only)
    const int voff0 = (4 * jg * D + 4 * d4) * 4;
    auto gload = [&](float4 (&rg)[4], int d0) {
        const int vb = (d0 + 4 * d4 < D) ? voff0 : 0x7ffffff0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(zr, vb + rr * D * 4, d0 * 4, (POL & 2) ? 2 : 0);
            rg[rr] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    const int nslab = (D + BD - 1) / BD;
    float4 r0[4], r1[4];
    gload(r0, 0);
    if constexpr (PF == 2) gload(r1, BD);

    f16x8 ah[KS], am[KS];
    {
        float* wl = reinterpret_cast<float*>(lds);
        const int nn = N * N;
        DKT_LDS_STAGE_OLD_LOOP(for (int i = tid; i < nn; i += NTH) wl[i] = Wb[i];)
        {
            LdsStage<NTH, NT> wst;                      // all of W in flight at once (dkt_split.h)
            wst.load(Wb, nn, tid);
            wst.store(wl, nn, tid);
        }
        __syncthreads();
        const int row = wave * 16 + r16;
        float v[KS][8];
        float rmax = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 32 * ks + 8 * q + e;
                v[ks][e] = (row < N && k < N) ? s * (wl[row * N + k] + wl[k * N + row]) : 0.f;
                rmax = fmaxf(rmax, fabsf(v[ks][e]));
            }
        }
        rmax = fmaxf(rmax, __shfl_xor(rmax, 16, DKT_WAVE));
        rmax = fmaxf(rmax, __shfl_xor(rmax, 32, DKT_WAVE));
        // power-of-two row scale: row maximum -> [2^14, 2^15); clamped so that its inverse (times 2^-15) stays normal
        const int eb = (int)((__float_as_uint(rmax) >> 23) & 0xffu);
        const int sexp = min(268 - eb, 237);
        const float rscale = __uint_as_float((unsigned)sexp << 23);
        if (q == 0) rowinv[row] = __uint_as_float((unsigned)(254 - sexp - 15) << 23);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xs = v[ks][e] * rscale;
                const _Float16 hi = (_Float16)xs;
                ah[ks][e] = hi;
                am[ks][e] = (_Float16)(xs - (float)hi);
            }
        }
        __syncthreads();
    }
    if constexpr (KP > NP) {
        constexpr int PADV = (KP - NP) / 8;
        for (int i = tid; i < NBUF * 2 * BD * PADV; i += NTH) {
            const int rowi = i / PADV, pc = i % PADV;
            _Float16* dst = zt + (size_t)rowi * RS + NP + 8 * pc;
            *reinterpret_cast<float4*>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    auto lstore = [&](const float4 (&rg)[4], int buf) {
        const float x[4][4] = {{rg[0].x, rg[1].x, rg[2].x, rg[3].x}, {rg[0].y, rg[1].y, rg[2].y, rg[3].y},
                               {rg[0].z, rg[1].z, rg[2].z, rg[3].z}, {rg[0].w, rg[1].w, rg[2].w, rg[3].w}};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f16x4 h, m;
            split2h(make_float4(x[t][0], x[t][1], x[t][2], x[t][3]), DKT_F16_SCALE, h, m);
            _Float16* dst = zt + (size_t)buf * 2 * PLANE + (16 * t + d4) * RS + 4 * jg;
            *reinterpret_cast<f16x4*>(dst) = h;
            *reinterpret_cast<f16x4*>(dst + PLANE) = m;
        }
    };
    const int st_row0 = wave * 16 + 4 * q;               // first of this lane's four output rows
    auto compute_store = [&](int buf, int d0) {
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const _Float16* base = zt + (size_t)buf * 2 * PLANE + r16 * RS + 8 * q;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const _Float16* p = base + 16 * t * RS + 32 * ks;
                const f16x8 bh = *reinterpret_cast<const f16x8*>(p);
                const f16x8 bm = *reinterpret_cast<const f16x8*>(p + PLANE);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bm, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am[ks], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bh, acc[t], 0, 0, 0);
            }
        }
        const int d = d0 + 4 * r16;
#ifndef DKT_GRAM_BWD_BRANCHY_STORES
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int row = wave * 16 + 4 * q + reg;
            const float u = rowinv[row];
#ifdef DKT_GRAM_BWD_BRANCHY_STORES        // the round-2..4 form: a global store inside an exec-masked block per row group (A/B builds, tools/gram_bwd_lib_ab.py)
            if (row < N && d < D) {
                const f32x4 o = {acc[0][reg] * u, acc[1][reg] * u, acc[2][reg] * u, acc[3][reg] * u};
                f32x4* dst = reinterpret_cast<f32x4*>(dZb + (size_t)row * D + d);
                if constexpr (POL & 1) __builtin_nontemporal_store(o, dst);
                else *dst = o;
            }
#else
            // Branch-free (round 5): rows / features past the end get an out-of-range offset.  Behind a branch hipcc cannot know how many stores are
            // outstanding when the NEXT image's registers are waited for and counts none -- `vmcnt(4)` for the last of them, which with the four stores
            // really in flight also drains the four FAR loads issued at the top of the stage: the second prefetch stage never stayed in flight.
            // (soffset stays the literal 0: see bstore4 in dkt_mfma_tiles.h.)
            // One live offset register: rows past N fall behind the descriptor's end by themselves ((row D + d) 4 >= 4 N D), only a feature past D needs the mask.
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 o = {__float_as_uint(acc[0][reg] * u), __float_as_uint(acc[1][reg] * u), __float_as_uint(acc[2][reg] * u), __float_as_uint(acc[3][reg] * u)};
            const int so = (d < D) ? (st_row0 + reg) * D * 4 + d * 4 : 0x7ffffff0;
            __builtin_amdgcn_raw_buffer_store_b128(o, dzr, so, 0, (POL & 1) ? 2 : 0);
#endif
        }
#ifndef DKT_GRAM_BWD_BRANCHY_STORES
        __builtin_amdgcn_sched_barrier(0);               // (the exec-masked blocks used to keep the staging code's live ranges out of this phase: 128 VGPRs = 2 workgroups per CU)
#endif
    };
    auto stage = [&](float4 (&rnear)[4], float4 (&rfar)[4], int sl) {
        const int buf = (NBUF == 2) ? (sl & 1) : 0;
        if constexpr (PF == 2) {
            gload(rfar, (sl + 2) * BD);
        } else {
            if (sl + 1 < nslab) gload(rnear, (sl + 1) * BD);
        }
        compute_store(buf, sl * BD);
        if constexpr (NBUF == 1) __syncthreads();
        if constexpr (PF == 2) {
            lstore(rnear, (NBUF == 2) ? (buf ^ 1) : 0);
        } else {
            if (sl + 1 < nslab) lstore(rnear, (NBUF == 2) ? (buf ^ 1) : 0);
        }
        __syncthreads();
    };
    __syncthreads();
    lstore(r0, 0);
    __syncthreads();
    if constexpr (PF == 2) {
        for (int sl = 0; sl < nslab; sl += 2) {
            stage(r1, r0, sl);
            stage(r0, r1, sl + 1);
        }
    } else {
        for (int sl = 0; sl < nslab; ++sl) stage(r0, r1, sl);
    }
}

// Measurement / validation switches: read from the environment ONCE, at the first launch -- not per call.
// dkt_reload_env() (below, exported for the test-suite and the A/B tools, which flip switches inside one process) re-reads them.
struct GramEnv {
    int ep, fewep, minb, split, ep_bk, ep_bd, unit_var, split_var, bwd_unit_var, bwd_split_var, bwd_unit_mind, bwd_split_mind;
    static int get(const char* name, int dflt) {                 // variant switch: twins library only
        const char* v = dkt_variant_env(name);
        return v ? atoi(v) : dflt;
    }
    static int get_product(const char* name, int dflt) {         // dispatch threshold: product library too
        const char* v = getenv(name);
        return v ? atoi(v) : dflt;
    }
    void load() {
        ep = get("DKT_GRAM_EP", 1); fewep = get("DKT_GRAM_FEWEP", 1); minb = get_product("DKT_GRAM_EP_MINB", 32); split = get("DKT_GRAM_SPLIT", 1);
        ep_bk = get("DKT_GRAM_EP_BK", 64); ep_bd = get("DKT_GRAM_EP_BD", 32);
        // Round 4 (tools/sweep_ep_variants.py + the in-step A/B of tools/r4_run12.sh, profiles/r04/v14_ep_variant_sweep.log): the forward with non-temporal Z loads
        // (22232: -1 % at D = 1600, -5 % at D = 512, -4 % at D = 64 against 2223) and, below D = 1024, the backward with ONE LDS image and prefetch depth 1
        // (211: 0.20 vs 0.24 ms at cfg1, 0.57 vs 0.68 ms at cfg3; equal at D = 1600, where 1222 -- two images, non-temporal dZ stores -- stays): 0 = this choice by D.
        unit_var = get("DKT_GRAM_UNIT_VAR", 22232); split_var = get("DKT_GRAM_SPLIT_VAR", 11);
        bwd_unit_var = get("DKT_GRAM_BWD_UNIT_VAR", 0); bwd_split_var = get("DKT_GRAM_BWD_SPLIT_VAR", 11);
        bwd_unit_mind = get("DKT_GRAM_BWD_UNIT_MIND", 64); bwd_split_mind = get("DKT_GRAM_BWD_SPLIT_MIND", 1024);
        lds_stage_env_sync();
    }
};
GramEnv& gram_env() {
    static GramEnv e = [] { GramEnv x; x.load(); return x; }();
    return e;
}

template <int NT>
void launch_sym(const float* Z, float* E, int B, int N, int D, int bk, bool unit, hipStream_t st) {
    if (bk == 3) {
        // <LDS buffers><prefetch depth> of the bf16 split; 2xxx = scaled-f16 split (unit-norm rows only):
        // 2223 = 2 stage buffers, prefetch depth 2, 3 workgroups per CU
        const int v = unit ? gram_env().unit_var : gram_env().split_var;
#ifdef DKT_TWINS             // the pipeline variants the defaults were chosen from (A/B runs, bitwise-twin tests)
        if (v == 21) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 2, 1>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 611) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 1, 64>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 612) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 2, 64>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 22) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 2, 2>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 12) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 2>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 211) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 1, 32, 2>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 212) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 2, 32, 2>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 2611) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 1, 64, 2>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 26113) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 1, 64, 2, 3>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 26114) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 1, 64, 2, 4>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 26122) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 2, 64, 2, 2>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 2223) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 2, 2, 32, 2, 3>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 2213) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 2, 1, 32, 2, 3>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
        if (v == 2115) { hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 1, 32, 2, 5>), dim3(B), dim3(256), 0, st, Z, E, N, D); return; }
#endif
        // the product: unit rows -> scaled 2-way f16 split, 2 stage buffers, prefetch depth 2, 3 workgroups per CU, non-temporal Z loads (22232);
        // any rows -> exact 3-way bf16 split, 1 buffer, depth 1 (11)
        // (NT = 8 at TWO workgroups per CU: built for three it spills 24 registers into the slab loop, and every scratch reload is a `vmcnt(0)` that drains the prefetch)
        if (v == 22232) hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 2, 2, 32, 2, NT == 8 ? 2 : 3, 2>), dim3(B), dim3(256), dkt_lds_pad("DKT_PAD_GRAM_EP"), st, Z, E, N, D);
        else hipLaunchKernelGGL((gram_sym_ep_split_kernel<NT, 1, 1>), dim3(B), dim3(256), 0, st, Z, E, N, D);
        return;
    }
#ifdef DKT_TWINS             // DKT_GRAM_SPLIT=0: the exact-fp32 MFMA kernels (validation twins)
    if (bk == 32) hipLaunchKernelGGL((gram_sym_ep_kernel<NT, 32>), dim3(B), dim3(256), 0, st, Z, E, N, D);
    else hipLaunchKernelGGL((gram_sym_ep_kernel<NT, 64>), dim3(B), dim3(256), 0, st, Z, E, N, D);
#endif
}

template <int NT>
void launch_bwd(const float* W, const float* Z, float* dZ, int B, int N, int D, const float* sc, int bd, bool unit, hipStream_t st) {
    if (bd == 3) {
        // <LDS buffers><prefetch depth> of the bf16 split; 2xx = scaled-f16 split (unit-norm rows of Z only)
        int v = unit ? gram_env().bwd_unit_var : gram_env().bwd_split_var;
        if (unit && v == 0) v = D < 1024 ? 211 : 1222;
#ifdef DKT_TWINS
        if (v == 222) { hipLaunchKernelGGL((gram_bwd_ep_f16x2_kernel<NT, 2, 2>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc); return; }
        if (v == 2222) { hipLaunchKernelGGL((gram_bwd_ep_f16x2_kernel<NT, 2, 2, 2>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc); return; }
        if (v == 3222) { hipLaunchKernelGGL((gram_bwd_ep_f16x2_kernel<NT, 2, 2, 3>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc); return; }
        if (v == 221) { hipLaunchKernelGGL((gram_bwd_ep_f16x2_kernel<NT, 2, 1>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc); return; }
        if (v == 212) { hipLaunchKernelGGL((gram_bwd_ep_f16x2_kernel<NT, 1, 2>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc); return; }
        if constexpr (NT <= 7) {
            if (v == 12) { hipLaunchKernelGGL((gram_bwd_ep_bf16x3_kernel<NT, 1, 2>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc); return; }
        }
#endif
        // the product: unit rows -> f16 split, one LDS image below D = 1024 (211), two + non-temporal dZ stores from there (1222); any rows -> bf16 split (11)
        if (v == 1222) { hipLaunchKernelGGL((gram_bwd_ep_f16x2_kernel<NT, 2, 2, 1>), dim3(B), dim3(64 * NT), dkt_lds_pad("DKT_PAD_GRAM_EP_BWD"), st, W, Z, dZ, N, D, sc); return; }
        if (v == 211) { hipLaunchKernelGGL((gram_bwd_ep_f16x2_kernel<NT, 1, 1>), dim3(B), dim3(64 * NT), dkt_lds_pad("DKT_PAD_GRAM_EP_BWD"), st, W, Z, dZ, N, D, sc); return; }
        if constexpr (NT <= 7) {                              // one stage buffer must also hold the N x N staging copy of W
            if (v == 11) { hipLaunchKernelGGL((gram_bwd_ep_bf16x3_kernel<NT, 1, 1>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc); return; }
        }
        hipLaunchKernelGGL((gram_bwd_ep_bf16x3_kernel<NT, 2, 2>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc);
        return;
    }
#ifdef DKT_TWINS
    if (bd != 32) { hipLaunchKernelGGL((gram_bwd_ep_kernel<NT, 64>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc); return; }
#endif
    hipLaunchKernelGGL((gram_bwd_ep_kernel<NT, 32>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc);      // short feature rows: exact-fp32 MFMA, 32-wide slabs
}

}  // namespace

// Returns true when the episode-resident kernel was launched.
bool dkt_gram_sym_ep_launch(const float* Z, float* E, int B, int N, int D, bool unit, hipStream_t st) {
    if (N <= 64 || N > 128 || (D & 3) || ((uintptr_t)Z & 15) || gram_env().ep == 0) return false;
    if (B < gram_env().minb) return false;
    // DKT_GRAM_SPLIT=1 (default): 3-way bf16 split on the bf16 MFMA pipe; 0: exact-fp32 MFMA (BK from DKT_GRAM_EP_BK)
    const int bk = gram_env().split ? 3 : gram_env().ep_bk;
    switch ((N + 15) / 16) {
        case 5: launch_sym<5>(Z, E, B, N, D, bk, unit, st); return true;
        case 6: launch_sym<6>(Z, E, B, N, D, bk, unit, st); return true;
        case 7: launch_sym<7>(Z, E, B, N, D, bk, unit, st); return true;
        case 8: launch_sym<8>(Z, E, B, N, D, bk, unit, st); return true;
        default: return false;
    }
}

// Fewer episodes than the episode-resident kernels take (B < DKT_GRAM_EP_MINB = 32): gram_sym_fewep_kernel (dkt_gram.hip), one workgroup per output tile, while
// those fit the chip at once; the backward stays with the generic tile kernel there.  The threshold is measured with the whole training step (105 x 1600) replayed
// from a hipGraph (tools/b1_minb_probe.sh, profiles/r06/b1_probe.log): 1 episode 0.086 ms (the generic forward kernel: 0.143), 8: 0.113 against 0.158 with the
// episode-resident pair, 16: 0.126 / 0.159, 24: 0.143 / 0.160, 32: 0.157 / 0.162.  (Until round 6 the threshold was 64 and the generic kernels served below it:
// 16 episodes 0.209 ms, 48: 0.245 ms.)
bool dkt_gram_fewep_applies(int B, int N, int D) {
    // (N <= 128: the range of the episode-resident kernels it stands in for; larger episodes keep the 64 x 64-tile kernels of dkt_gram_big.hip at every batch)
    return gram_env().fewep != 0 && (D & 3) == 0 && N <= 128 && B < gram_env().minb;
}

bool dkt_gram_bwd_ep_launch(const float* W, const float* Z, float* dZ, int B, int N, int D, const float* sc, bool unit, hipStream_t st) {
    if (N <= 64 || N > 128 || (D & 3) || ((uintptr_t)Z & 15) || ((uintptr_t)dZ & 15) || gram_env().ep == 0) return false;
    if (B < gram_env().minb) return false;
    // the split kernel pays a per-episode setup (A-fragment split, LDS zero fill): it wins from ~16 slabs of 64 features
    // (the f16 kernel for unit-norm rows has the cheaper staging path and already wins at D = 64: 0.27 vs 0.40 ms per 8192 episodes)
    const int mind = unit ? gram_env().bwd_unit_mind : gram_env().bwd_split_mind;
    const int bd = (gram_env().split && D >= mind) ? 3 : gram_env().ep_bd;
    switch ((N + 15) / 16) {
        case 5: launch_bwd<5>(W, Z, dZ, B, N, D, sc, bd, unit, st); return true;
        case 6: launch_bwd<6>(W, Z, dZ, B, N, D, sc, bd, unit, st); return true;
        case 7: launch_bwd<7>(W, Z, dZ, B, N, D, sc, bd, unit, st); return true;
        case 8: launch_bwd<8>(W, Z, dZ, B, N, D, sc, bd, unit, st); return true;
        default: return false;
    }
}

// Re-read the measurement switches (tests / A-B tools flip them inside one process).  Not part of include/dkt_abi.h.
void dkt_mll_h2_reload_env();          // dkt_mll_h2.hip: DKT_MLL_H2E_MINB
void dkt_mll_tiled_reload_env();       // dkt_mll_tiled.hip: DKT_MLL_TILED_F16
void dkt_frontend_reload_env();         // dkt_frontend.hip: DKT_GRAM_EP_MINB / DKT_GRAM_DIST_EP of the episode-resident distance build
void dkt_mll_reload_env();              // dkt_mll.hip: DKT_MLL_F32MFMA, DKT_MLL_P2_GUARD
void dkt_gram_big_reload_env();         // dkt_gram_big.hip: DKT_GRAM_BIG_EP
void dkt_classkernel_reload_env();      // dkt_classkernel.hip: DKT_CLASS_BWD_V4
void dkt_gram_small_reload_env();       // dkt_gram_small.hip: DKT_GRAM_SMALL_WG
extern "C" void dkt_reload_env(void) {
    gram_env().load(); dkt_mll_h2_reload_env(); dkt_mll_tiled_reload_env(); dkt_frontend_reload_env(); dkt_mll_reload_env(); dkt_gram_big_reload_env();
    dkt_classkernel_reload_env(); dkt_gram_small_reload_env();
}

// DKT_GRAM_SPLIT (default 1): the split-precision kernels may be used (0: exact-fp32 MFMA kernels everywhere) -- for dkt_gram_big.hip
bool dkt_gram_split_enabled() { return gram_env().split != 0; }
