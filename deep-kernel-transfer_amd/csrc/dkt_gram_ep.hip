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

template <int ROW>
__device__ __forceinline__ void sym_store_row(const f32x4* acc, float* Eb, int N, int r16, int q) {
#pragma unroll
    for (int tj = 0; tj <= ROW; ++tj) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int gi = ROW * 16 + 4 * q + reg, gj = tj * 16 + r16;
            if (gi < N && gj < N && gj <= gi) {
                const float v = acc[tj][reg];
                Eb[gi * N + gj] = v;
                if (gi != gj) Eb[gj * N + gi] = v;
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

template <int NT>
void launch_sym(const float* Z, float* E, int B, int N, int D, int bk, hipStream_t st) {
    if (bk == 32) hipLaunchKernelGGL((gram_sym_ep_kernel<NT, 32>), dim3(B), dim3(256), 0, st, Z, E, N, D);
    else hipLaunchKernelGGL((gram_sym_ep_kernel<NT, 64>), dim3(B), dim3(256), 0, st, Z, E, N, D);
}

template <int NT>
void launch_bwd(const float* W, const float* Z, float* dZ, int B, int N, int D, const float* sc, int bd, hipStream_t st) {
    if (bd == 32) hipLaunchKernelGGL((gram_bwd_ep_kernel<NT, 32>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc);
    else hipLaunchKernelGGL((gram_bwd_ep_kernel<NT, 64>), dim3(B), dim3(64 * NT), 0, st, W, Z, dZ, N, D, sc);
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

}  // namespace

// Returns true when the episode-resident kernel was launched.
bool dkt_gram_sym_ep_launch(const float* Z, float* E, int B, int N, int D, hipStream_t st) {
    if (N <= 64 || N > 128 || (D & 3) || ((uintptr_t)Z & 15) || env_int("DKT_GRAM_EP", 1) == 0) return false;
    if (B < env_int("DKT_GRAM_EP_MINB", 64)) return false;
    const int bk = env_int("DKT_GRAM_EP_BK", 64);
    switch ((N + 15) / 16) {
        case 5: launch_sym<5>(Z, E, B, N, D, bk, st); return true;
        case 6: launch_sym<6>(Z, E, B, N, D, bk, st); return true;
        case 7: launch_sym<7>(Z, E, B, N, D, bk, st); return true;
        case 8: launch_sym<8>(Z, E, B, N, D, bk, st); return true;
        default: return false;
    }
}

bool dkt_gram_bwd_ep_launch(const float* W, const float* Z, float* dZ, int B, int N, int D, const float* sc, hipStream_t st) {
    if (N <= 64 || N > 128 || (D & 3) || ((uintptr_t)Z & 15) || ((uintptr_t)dZ & 15) || env_int("DKT_GRAM_EP", 1) == 0) return false;
    if (B < env_int("DKT_GRAM_EP_MINB", 64)) return false;
    const int bd = env_int("DKT_GRAM_EP_BD", 32);
    switch ((N + 15) / 16) {
        case 5: launch_bwd<5>(W, Z, dZ, B, N, D, sc, bd, st); return true;
        case 6: launch_bwd<6>(W, Z, dZ, B, N, D, sc, bd, st); return true;
        case 7: launch_bwd<7>(W, Z, dZ, B, N, D, sc, bd, st); return true;
        case 8: launch_bwd<8>(W, Z, dZ, B, N, D, sc, bd, st); return true;
        default: return false;
    }
}
