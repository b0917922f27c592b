// dkt_frontend.hip -- the BNCosSim front half of the deep kernel fused into the Gram build:
//   reference  z = trunk(x) ending in bn_out = BatchNorm1d(D)   (methods/DKT.py:48)
//              z = F.normalize(z, p=2, dim=1)                   (methods/DKT.py:141-142, 174-175, 236-237)
//              K = LinearKernel(z, z)                           (methods/DKT.py:375-378)
// here       dkt_bn_stats_f32   : per-episode batch statistics of the raw trunk output X[b] (train mode), folded into
//                                 an affine map  y = a x + s   (a = gamma rstd, s = beta - mean a)
//            dkt_gram_bn_f32    : E = Zn Zn^T with Zn_i = y_i / max(||y_i||, 1e-12) -- the affine map is applied while the
//                                 slice is staged, the row norms come out of the diagonal of G' = Y Y^T, Zn is never written
// (dkt_gram_bn_bwd_f32, the matching backward, lives below.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

#include "dkt_common.h"
#include "dkt_tiles.h"
#include "dkt_split.h"
#include "../../include/dkt_abi.h"

namespace {

typedef __amdgpu_buffer_rsrc_t brsrc_t;
__device__ __forceinline__ brsrc_t mk_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 bload4(brsrc_t r, int voff, int soff) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
constexpr int OOB = 0x7ffffff0;
__device__ __forceinline__ void fe_store4(brsrc_t r, int voff, float a0, float a1, float a2, float a3) {     // soffset = literal 0: see bstore4, dkt_mfma_tiles.h
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {__float_as_uint(a0), __float_as_uint(a1), __float_as_uint(a2), __float_as_uint(a3)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Batch statistics: grid (ceil(D / 256), B), 1024 threads = 64 feature quads x 16 row lanes; a row lane walks the rows rl, rl + 16, ... of its quad
// (float4 loads, 1 KB contiguous per wave and row, 8 rows in flight), the 16 lanes of a quad meet in a fixed tree in LDS (bitwise reproducible).  Sums are
// taken about the first row (shifted data): no catastrophic cancellation in  E[x^2] - E[x]^2  for features with a large common offset (ReLU outputs).
// (Until round 6 a thread walked all N rows of its quad: 128 busy threads per 420 x 512 episode, 0.25 of the HBM roofline at the 20-way shape.)
__global__ __launch_bounds__(1024) void bn_stats_kernel(const float* __restrict__ X, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, float* __restrict__ mean,
                                                        float* __restrict__ rstd, float* __restrict__ a, float* __restrict__ s,
                                                        float* __restrict__ var_unbiased, int N, int D) {
    __shared__ __attribute__((aligned(16))) float red[16][64][8];
    const int b = blockIdx.y;
    const int qd = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int d = 256 * blockIdx.x + 4 * qd;
    const bool dok = d < D;                                     // (D % 4 == 0: a quad is inside the row or wholly beyond it)
    const float* Xb = X + (size_t)b * N * D + (dok ? d : 0);
    const float4 x0 = *reinterpret_cast<const float4*>(Xb);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (dok) {
#pragma unroll 8
        for (int i = rl; i < N; i += 16) {                      // (row 0 contributes exact zeros)
            const float4 v = *reinterpret_cast<const float4*>(Xb + (size_t)i * D);
            const float e[4] = {v.x - x0.x, v.y - x0.y, v.z - x0.z, v.w - x0.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s1[t] += e[t];
                s2[t] = __builtin_fmaf(e[t], e[t], s2[t]);
            }
        }
    }
    *reinterpret_cast<float4*>(&red[rl][qd][0]) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    *reinterpret_cast<float4*>(&red[rl][qd][4]) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    __syncthreads();
#pragma unroll
    for (int w = 8; w > 0; w >>= 1) {
        if (rl < w) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[rl][qd][e] += red[rl + w][qd][e];
        }
        __syncthreads();
    }
    if (rl != 0 || !dok) return;
#pragma unroll
    for (int t = 0; t < 4; ++t) { s1[t] = red[0][qd][t]; s2[t] = red[0][qd][4 + t]; }
    const float inv_n = 1.0f / (float)N;
    const float x0v[4] = {x0.x, x0.y, x0.z, x0.w};
    float mu[4], rs[4], av[4], sv[4], vu[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const float m1 = s1[t] * inv_n;
        const float var = fmaxf(__builtin_fmaf(-m1, m1, s2[t] * inv_n), 0.f);    // biased variance (normalisation)
        mu[t] = x0v[t] + m1;
        rs[t] = 1.0f / sqrtf(var + eps);
        const float g = gamma ? gamma[d + t] : 1.0f, be = beta ? beta[d + t] : 0.0f;
        av[t] = g * rs[t];
        sv[t] = __builtin_fmaf(-mu[t], av[t], be);
        vu[t] = (N > 1) ? var * (float)N / (float)(N - 1) : var;                  // what torch feeds the running variance
    }
    const size_t o = (size_t)b * D + d;
    *reinterpret_cast<float4*>(mean + o) = make_float4(mu[0], mu[1], mu[2], mu[3]);
    *reinterpret_cast<float4*>(rstd + o) = make_float4(rs[0], rs[1], rs[2], rs[3]);
    *reinterpret_cast<float4*>(a + o) = make_float4(av[0], av[1], av[2], av[3]);
    *reinterpret_cast<float4*>(s + o) = make_float4(sv[0], sv[1], sv[2], sv[3]);
    if (var_unbiased) *reinterpret_cast<float4*>(var_unbiased + o) = make_float4(vu[0], vu[1], vu[2], vu[3]);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct BnTrainOut {                // train-mode statistics written by the STATS variant of the fused forward (each [B, D])
    float *mean, *rstd, *a, *s, *var_unbiased;
    float eps;
    int has_gamma, has_beta;
    const float* lengthscale;      // distance epilogues: [1]
    int epi;                       // 0: cosine similarities (this front end); 1: d2 / l^2; 2: exp(-d2 / 2 l^2)  (dkt_gram_f32 kinds SQDIST / RBF)
};

// ---------------------------------------------------------------------------------------------
// Fused forward: one 256-thread workgroup per episode, the structure of gram_sym_ep_split_kernel<NT, 1, 1> (bf16 split).
// STATS (round 3; dkt_gram_bn_train_f32): the train-mode batch statistics are taken INSIDE the staging path -- all N rows of a
// 32-feature slice pass through the workgroup's registers at once, so the column sums (about row 0: shifted data, as
// bn_stats_kernel) are reduced over the 32 threads that hold a feature (three lane exchanges + a 1-KB table in LDS that rides on
// the slab loop's two existing barriers), folded into a = gamma rstd, s = beta - mean a, and applied to the same registers.
// X is read ONCE; A / S are then gamma / beta [D].
// EPI (round 3; dkt_gram_f32 kinds SQDIST / RBF at 32 < N <= 128): the same kernel as the episode-resident squared-distance build -- the
// affine map is y = x - x_0 (row 0 of the episode: the shift GPyTorch's mean-centring provides, taken while the slice is staged), the
// epilogue reads |y_i|^2 off the diagonal and emits d2_ij / l^2 = (|y_i|^2 + |y_j|^2 - 2 y_i . y_j) / l^2 (EPI = 1) or exp(-d2 / 2 l^2)
// (EPI = 2), exact zero / unit diagonal, bitwise symmetric.  A / S are unused.
// redo_only (round 5): the fix-up pass behind gram_bn_train_f16_kernel (below) -- workgroups whose episode that kernel did not flag (rnorm[b, 0] is not
// NaN) exit at once; a flagged episode is computed here in full, in the 3-way bf16 split that needs no bound on the operands.
template <int NT, bool STATS, int EPI = 0>
__global__ __launch_bounds__(256, NT <= 6 ? 4 : (NT == 7 ? 3 : 2)) void gram_bn_sym_ep_kernel(const float* __restrict__ X, const float* __restrict__ A,
                                                                              const float* __restrict__ S, long ab_bstride,
                                                                              float* __restrict__ E, float* __restrict__ rnorm,
                                                                              int N, int D, BnTrainOut bo, int redo_only) {
    constexpr int NP = 16 * NT;
    constexpr int BK = 32;
    constexpr int SPLD = BK + 16;
    constexpr int V4_PER_ROW = BK / 4;
    constexpr int NV4 = NP * V4_PER_ROW;
    constexpr int NLD = (NV4 + 255) / 256;
    constexpr int NPL = NLD * 256 / V4_PER_ROW;
    constexpr int PLANE = NPL * SPLD;
    __shared__ __attribute__((aligned(16))) __bf16 zp[3 * PLANE];
    __shared__ float rho[NP];
    __shared__ __attribute__((aligned(16))) float red[STATS ? 4 * V4_PER_ROW * 8 : 4];     // [wave][c4][s1 x 4, s2 x 4]
    __shared__ __attribute__((aligned(16))) float redx0[STATS ? BK : 4];                   // row 0 of the slice: the shift of the sums
    __shared__ __attribute__((aligned(16))) float fold_as[STATS ? 4 * 2 * BK : 4];         // [wave][a x 32, s x 32]: wave-private

    const int b = blockIdx.x;
    if (redo_only) {                                     // fix-up pass behind the f16 instance: only the episodes it flagged
        const float f = rnorm[(size_t)b * N];
        if (f == f) return;
    }
    float* Eb = E + (size_t)b * N * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const brsrc_t xr = mk_rsrc(X + (size_t)b * N * D, N * D * 4);
    const brsrc_t ar = mk_rsrc(A + (size_t)b * ab_bstride, D * 4);
    const brsrc_t sr = mk_rsrc(S + (size_t)b * ab_bstride, D * 4);
    const int c4 = tid % V4_PER_ROW;                     // the thread's 4 features of every slice (same for all its rows)
    int voff[NLD];
    bool rowok[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int row = (tid + 256 * i) / V4_PER_ROW;
        rowok[i] = row < N;
        voff[i] = rowok[i] ? (row * D + 4 * c4) * 4 : OOB;
    }
    float4 rg[NLD], av, sv, x0;
    f32x4 acc[NT + 1];
#pragma unroll
    for (int i = 0; i <= NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float gsc = 1.0f, bsc = 0.0f;                        // STATS: gamma / beta of feature k0 + lane % 32
    int kcur = 0;                                        // first feature of the slice held in rg (STATS: where its statistics go)
    auto gload = [&](int k0) {
        const bool in = k0 + 4 * c4 < D;                 // ragged last slice: features past D read as zeros (a = s = 0 too)
        if constexpr (STATS) {
            const int f = lane & (BK - 1);
            const int fo = (k0 + f < D) ? 4 * f : OOB;
            if (bo.has_gamma) gsc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ar, fo, k0 * 4, 0));
            if (bo.has_beta) bsc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sr, fo, k0 * 4, 0));
            x0 = bload4(xr, in ? 16 * c4 : OOB, k0 * 4); // row 0 of the thread's features: the shift of the sums
            kcur = k0;
        } else if constexpr (EPI != 0) {
            x0 = bload4(xr, in ? 16 * c4 : OOB, k0 * 4);
            av = make_float4(1.f, 1.f, 1.f, 1.f);
            sv = make_float4(-x0.x, -x0.y, -x0.z, -x0.w);
        } else {
            av = bload4(ar, in ? 16 * c4 : OOB, k0 * 4);
            sv = bload4(sr, in ? 16 * c4 : OOB, k0 * 4);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) rg[i] = bload4(xr, in ? voff[i] : OOB, k0 * 4);
    };
    // STATS, step 1 (before the barrier): the thread's partial sums about row 0 (packed fp32 math), reduced over the 8 lanes of the wave
    // that hold the same features with a reduce-scatter (row_ror:8, then v_permlane32_swap / v_permlane16_swap on PAIRS of values: one
    // swap + one add halves the number of live values), published per wave
    auto stats_partial = [&]() {
        const f32x2 xa = {x0.x, x0.y}, xb = {x0.z, x0.w};
        f32x2 s1a = {0.f, 0.f}, s1b = s1a, s2a = s1a, s2b = s1a;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            // padded rows were loaded as zeros (e = -x0): multiplied out exactly (a subtract-afterwards correction leaves the rounding
            // residue of x0^2 per padded row in the sum of squares -- measured 1e-4 relative in the variance at N = 19)
            const f32x2 mk = {rowok[i] ? 1.0f : 0.0f, rowok[i] ? 1.0f : 0.0f};
            const f32x2 ea = ((f32x2){rg[i].x, rg[i].y} - xa) * mk, eb = ((f32x2){rg[i].z, rg[i].w} - xb) * mk;
            s1a += ea;
            s1b += eb;
            s2a += ea * ea;
            s2b += eb * eb;
        }
        float v[8] = {s1a.x, s1a.y, s1b.x, s1b.y, s2a.x, s2a.y, s2b.x, s2b.y};
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[t]), 0x128, 0xf, 0xf, false));   // lane ^ 8
        float u[4], w2[2];
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {                 // lanes 0..31: v[2 p2] summed over lane ^ 32, lanes 32..63: v[2 p2 + 1]
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * p2]), __float_as_uint(v[2 * p2 + 1]), false, false);
            u[p2] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {                 // 16-lane rows: [u[2 p2] lower, u[2 p2 + 1] lower, u[2 p2] upper, u[2 p2 + 1] upper] completed
            const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[2 * p2]), __float_as_uint(u[2 * p2 + 1]), false, false);
            w2[p2] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
        // row r of w2[0] holds v[{0, 2, 1, 3}[r]] (the s1 of feature ...), of w2[1] v[4 + ...] (the s2)
        if ((lane & 8) == 0) {
            const int r = lane >> 4, idx = ((r & 1) << 1) | (r >> 1);
            red[(wave * V4_PER_ROW + c4) * 8 + idx] = w2[0];
            red[(wave * V4_PER_ROW + c4) * 8 + 4 + idx] = w2[1];
        }
        if (tid < V4_PER_ROW) *reinterpret_cast<float4*>(&redx0[4 * c4]) = x0;
    };
    // STATS, step 2 (after the barrier): lane f % 32 of every wave folds feature f of the slice (the four waves do the same work: no
    // second barrier), the a / s of the slice go through a wave-private table back to the threads that stage those features
    auto stats_fold = [&]() {
        const int f = lane & (BK - 1), fc = f >> 2, ft = f & 3;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            s1 += red[(w * V4_PER_ROW + fc) * 8 + ft];
            s2 += red[(w * V4_PER_ROW + fc) * 8 + 4 + ft];
        }
        const float inv_n = 1.0f / (float)N;
        const float m1 = s1 * inv_n;
        const float var = fmaxf(__builtin_fmaf(-m1, m1, s2 * inv_n), 0.f);      // biased variance (normalisation)
        const float mu = redx0[f] + m1;
        const float ve = var + bo.eps;
        float rs = __builtin_amdgcn_rsqf(ve);
        rs = rs * __builtin_fmaf(-0.5f * ve * rs, rs, 1.5f);                     // one Newton step: v_rsq_f32 is ~1 ulp
        const float aa = gsc * rs;
        const float ss = __builtin_fmaf(-mu, aa, bsc);
        float* tab = &fold_as[wave * 2 * BK];
        tab[f] = aa;
        tab[BK + f] = ss;
        if (wave == 0 && lane < BK && kcur + f < D) {
            const size_t o = (size_t)b * D + kcur + f;
            bo.mean[o] = mu;
            bo.rstd[o] = rs;
            bo.a[o] = aa;
            bo.s[o] = ss;
            if (bo.var_unbiased) bo.var_unbiased[o] = (N > 1) ? var * (float)N / (float)(N - 1) : var;      // what torch feeds the running variance
        }
        av = *reinterpret_cast<const float4*>(&tab[4 * c4]);
        sv = *reinterpret_cast<const float4*>(&tab[BK + 4 * c4]);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = (tid + 256 * i) / V4_PER_ROW;
            float4 y;
            y.x = rowok[i] ? __builtin_fmaf(av.x, rg[i].x, sv.x) : 0.f;      // padded rows must stay 0 (their y would be s)
            y.y = rowok[i] ? __builtin_fmaf(av.y, rg[i].y, sv.y) : 0.f;
            y.z = rowok[i] ? __builtin_fmaf(av.z, rg[i].z, sv.z) : 0.f;
            y.w = rowok[i] ? __builtin_fmaf(av.w, rg[i].w, sv.w) : 0.f;
            bf16x4 h, m, l;
            split3(y, h, m, l);
            __bf16* dst = &zp[row * SPLD + 4 * c4];
            *reinterpret_cast<bf16x4*>(dst) = h;
            *reinterpret_cast<bf16x4*>(dst + PLANE) = m;
            *reinterpret_cast<bf16x4*>(dst + 2 * PLANE) = l;
        }
    };
    auto tiles = [&](auto rows) {
        using R = decltype(rows);
        if constexpr (R::RA >= 0) sym_tiles_mfma_bf16x3<NT, R::RA, R::RB, SPLD, PLANE>(acc, zp, r16, q);
    };

    const int nk = (D + BK - 1) / BK;
    gload(0);
    if constexpr (STATS) {
        stats_partial();
        __syncthreads();
        stats_fold();
    }
    lstore();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload((kt + 1) * BK);
        if (wave == 0) tiles(RowsOf<NT, 0>{});
        else if (wave == 1) tiles(RowsOf<NT, 1>{});
        else if (wave == 2) tiles(RowsOf<NT, 2>{});
        else tiles(RowsOf<NT, 3>{});
        if constexpr (STATS) {
            if (kt + 1 < nk) stats_partial();            // the table was last read before the previous barrier
        }
        __syncthreads();
        if (kt + 1 < nk) {
            if constexpr (STATS) stats_fold();
            lstore();
        }
        __syncthreads();
    }

    // ---- row norms from the diagonal of G' = Y Y^T, then E_ij = G'_ij rho_i rho_j ----
    auto put_diag = [&](const f32x4& t, int row_blk) {   // diagonal tile (row_blk, row_blk): lane (r16, q) holds rows 4q+reg, col r16
        if ((r16 >> 2) == q) {
            const int rr = r16 & 3;
            const float v = rr == 0 ? t[0] : rr == 1 ? t[1] : rr == 2 ? t[2] : t[3];
            if constexpr (EPI != 0) rho[16 * row_blk + r16] = fmaxf(v, 0.f);          // |y_i|^2
            else rho[16 * row_blk + r16] = 1.0f / fmaxf(sqrtf(fmaxf(v, 0.f)), 1e-12f);    // F.normalize: x / max(||x||, 1e-12)
        }
    };
    auto diag_of = [&](auto w) {
        constexpr int W = decltype(w)::value;
        constexpr int RA = RowsOf<NT, W>::RA, RB = RowsOf<NT, W>::RB;
        if constexpr (RA >= 0) put_diag(acc[RA], RA);
        if constexpr (RB >= 0) put_diag(acc[RA + 1 + RB], RB);
    };
    if (wave == 0) diag_of(std::integral_constant<int, 0>{});
    else if (wave == 1) diag_of(std::integral_constant<int, 1>{});
    else if (wave == 2) diag_of(std::integral_constant<int, 2>{});
    else diag_of(std::integral_constant<int, 3>{});
    __syncthreads();
    if constexpr (EPI == 0) {
        if (tid < N) rnorm[(size_t)b * N + tid] = rho[tid];
    }
    float inv_l2 = 0.f;
    if constexpr (EPI != 0) {
        const float l = bo.lengthscale[0];
        inv_l2 = 1.0f / (l * l);
    }
    auto emit = [&](f32x4& t, const int rowblk, const int tj) {     // tile (rowblk, tj) of G' -> the kernel's output values, in place
        const float rj = rho[16 * tj + r16];
        const f32x4 ri = *reinterpret_cast<const f32x4*>(&rho[16 * rowblk + 4 * q]);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            if constexpr (EPI == 0) {
                t[reg] *= ri[reg] * rj;
            } else {
                float d2 = fmaxf(ri[reg] + rj - 2.0f * t[reg], 0.f);
                if (rowblk == tj && 4 * q + reg == r16) d2 = 0.f;
                t[reg] = (EPI == 2) ? expf(-0.5f * d2 * inv_l2) : d2 * inv_l2;
            }
        }
    };
    auto finish = [&](auto w) {
        constexpr int W = decltype(w)::value;
        constexpr int RA = RowsOf<NT, W>::RA, RB = RowsOf<NT, W>::RB;
        if constexpr (RA >= 0) {
#pragma unroll
            for (int tj = 0; tj <= RA; ++tj) emit(acc[tj], RA, tj);
            sym_store_row<RA>(acc, Eb, N, r16, q);
        }
        if constexpr (RB >= 0) {
#pragma unroll
            for (int tj = 0; tj <= RB; ++tj) emit(acc[RA + 1 + tj], RB, tj);
            sym_store_row<RB>(acc + RA + 1, Eb, N, r16, q);
        }
    };
    if (wave == 0) finish(std::integral_constant<int, 0>{});
    else if (wave == 1) finish(std::integral_constant<int, 1>{});
    else if (wave == 2) finish(std::integral_constant<int, 2>{});
    else finish(std::integral_constant<int, 3>{});
}

// ---------------------------------------------------------------------------------------------
// Train-mode fused forward on the f16 pipe (round 5; dkt_gram_bn_train_f32 at N > 32).
// Arithmetic: the scaled 2-way f16 split of the level-1 unit-row kernel instead of the 3-way bf16 split -- three MFMA products instead of six, two LDS
// planes instead of three, ~3 instead of 5.5 VALU instructions per staged element.  Rounds 2-4 kept bf16 here because y = a x + s is not bounded by 1.  But
// train-mode BatchNorm bounds every element a priori, |y_ik| <= |beta_k| + |gamma_k| sqrt(N - 1) (the largest z-score N samples can hold), so the power-of-two
// scale S that keeps the high piece inside f16 is known before the first slice (a max over gamma / beta in the prologue): no overflow is possible.  What the
// bound cannot give is the other side: the MFMA flushes f16 subnormals, i.e. a low piece below 2^-14 in scaled units, an absolute error of <= sqrt(D) 2^-14
// per ROW -- nothing against a row of typical norm (S |y_i| ~ 2^16 at D = 1600), not so for a row whose norm is orders of magnitude below the element bound.
// That is checked a posteriori on the diagonal of the scaled G' (S^2 |y_i|^2 >= D 2^14  <=>  flush error <= 2^-21 of the row norm; and finite): an episode
// that fails it gets NaN in rnorm[b, 0], writes no E, and is redone by gram_bn_sym_ep_kernel<NT, true> (bf16 x 3) in the fix-up launch right behind, where
// every other workgroup exits at once -- the pattern of the tile-array marginal likelihood; no host round trip.  Measured error against float64: 2.4e-7
// (bf16 x 3: 1.8e-7..2.4e-7; tools/fe_fwd_ab.py).
// Schedule: that of gram_sym_ep_split_kernel<NT, 2, 2, 32, 2> -- two LDS images, two register stages of raw slices in flight, ONE barrier per
// 32-feature slice (the round-3 kernel: one image, two barriers).  A stage (slice kt multiplied from image kt & 1):
//     fold the statistics of slice kt + 1 (published before the last barrier) -> a, s;  apply + split + store it into image (kt + 1) & 1
//     issue the loads of slice kt + 3 into the registers that just became free
//     the MFMA products of slice kt
//     partial column sums of slice kt + 2 (its loads were issued a whole stage ago) -> table kt & 1
//     barrier
// Every cross-wave dependence (image, partial-sum table, row-0 shift) is one barrier apart, written -> read or read -> overwritten.  The loop is
// branch-free (slices past D load as zeros through the descriptor, get a = s = 0 and multiply as zeros; an odd slice count is rounded up; absent gamma /
// beta are empty descriptors; the statistics leave through buffer stores with out-of-range offsets for idle lanes) so that the compiler's vmcnt
// bookkeeping is exact and the far loads really stay in flight across a stage.
// Measured (profiles/r05/v5_fe_fwd_ab.log, same box): 2048 cfg2 episodes 0.550 -> 0.369 ms, 8192: 1.858 -> 1.345 ms (0.41 -> 0.57 of 8 TB/s); the same
// split in the round-3 schedule: 0.469 / 1.603 ms.
template <int NT>
__global__ __launch_bounds__(256, NT <= 5 ? 4 : (NT <= 7 ? 3 : 2)) void gram_bn_train_f16_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                                                 const float* __restrict__ Bt, float* __restrict__ E,
                                                                                 float* __restrict__ rnorm, int N, int D, BnTrainOut bo) {
    constexpr int NP = 16 * NT;
    constexpr int BK = 32;
    constexpr int SPLD = BK + 16;
    constexpr int V4_PER_ROW = BK / 4;
    constexpr int NV4 = NP * V4_PER_ROW;
    constexpr int NLD = (NV4 + 255) / 256;
    constexpr int NPL = NLD * 256 / V4_PER_ROW;
    constexpr int PLANE = NPL * SPLD;
    __shared__ __attribute__((aligned(16))) _Float16 zp[2][2 * PLANE];
    __shared__ float rho[NP];
    __shared__ __attribute__((aligned(16))) float red[2][4 * V4_PER_ROW * 8];      // [table][wave][c4][s1 x 4, s2 x 4]
    __shared__ __attribute__((aligned(16))) float redx0[2][BK];                    // row 0 of the slice: the shift of the sums
    __shared__ __attribute__((aligned(16))) float fold_as[4 * 2 * BK];             // [wave][a x 32, s x 32]: wave-private
    __shared__ float bmax_w[4];
    __shared__ int bad;

    const int b = blockIdx.x;
    float* Eb = E + (size_t)b * N * N;
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const brsrc_t xr = mk_rsrc(X + (size_t)b * N * D, N * D * 4);
    // absent gamma / beta: an empty descriptor -- every load returns 0 (gamma = 0 + 1, beta = 0) with no branch around it
    const brsrc_t gr = mk_rsrc(G, bo.has_gamma ? D * 4 : 0);
    const brsrc_t br = mk_rsrc(Bt, bo.has_beta ? D * 4 : 0);
    const float gdef = bo.has_gamma ? 0.0f : 1.0f;
    // the statistics leave through buffer stores: lanes with nothing to store (waves 1-3, the upper half of wave 0, features past D) get an
    // out-of-range offset, a NULL var_unbiased an empty descriptor -- no exec-masked region, no branch inside the loop
    const size_t so = (size_t)b * D;
    const brsrc_t o_mean = mk_rsrc(bo.mean + so, D * 4), o_rstd = mk_rsrc(bo.rstd + so, D * 4), o_a = mk_rsrc(bo.a + so, D * 4),
                  o_s = mk_rsrc(bo.s + so, D * 4), o_var = mk_rsrc(bo.var_unbiased ? bo.var_unbiased + so : bo.mean, bo.var_unbiased ? D * 4 : 0);
    const bool st_lane = wave == 0 && lane < BK;
    const float unb = (N > 1) ? (float)N / (float)(N - 1) : 1.0f;      // biased -> unbiased variance: what torch feeds the running estimate
    const int c4 = tid % V4_PER_ROW;
    int voff[NLD];
    bool rowok[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int row = (tid + 256 * i) / V4_PER_ROW;
        rowok[i] = row < N;
        voff[i] = rowok[i] ? (row * D + 4 * c4) * 4 : OOB;
    }
    struct Slice {
        float4 rg[NLD], x0;
        float gsc, bsc;                                  // gamma / beta of feature k0 + lane % 32
    };
    auto gload = [&](Slice& r, int k0) {
        const bool in = k0 + 4 * c4 < D;                 // ragged last slice and the slices past D: zeros
        const int f = lane & (BK - 1);
        const int fo = (k0 + f < D) ? 4 * f : OOB;
        r.gsc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gr, fo, k0 * 4, 0));
        r.bsc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(br, fo, k0 * 4, 0));
        r.x0 = bload4(xr, in ? 16 * c4 : OOB, k0 * 4);
#pragma unroll
        for (int i = 0; i < NLD; ++i) r.rg[i] = bload4(xr, in ? voff[i] : OOB, k0 * 4);
    };
    // partial sums about row 0 (shifted data), reduced over the 8 lanes of the wave that hold the same features: as stats_partial above
    auto partial = [&](const Slice& r, const int tab) {
        const f32x2 xa = {r.x0.x, r.x0.y}, xb = {r.x0.z, r.x0.w};
        f32x2 s1a = {0.f, 0.f}, s1b = s1a, s2a = s1a, s2b = s1a;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            f32x2 ea = (f32x2){r.rg[i].x, r.rg[i].y} - xa, eb = (f32x2){r.rg[i].z, r.rg[i].w} - xb;
            if (i == NLD - 1) {                          // only the last 32-row group can hold padded rows (N > 32 (NLD - 1) for every NT >= 3): multiplied out exactly
                const f32x2 mk = {rowok[i] ? 1.0f : 0.0f, rowok[i] ? 1.0f : 0.0f};
                ea *= mk;
                eb *= mk;
            }
            s1a += ea;
            s1b += eb;
            s2a += ea * ea;
            s2b += eb * eb;
        }
        float v[8] = {s1a.x, s1a.y, s1b.x, s1b.y, s2a.x, s2a.y, s2b.x, s2b.y};
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[t]), 0x128, 0xf, 0xf, false));   // lane ^ 8
        float u[4], w2[2];
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
            const auto x = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * p2]), __float_as_uint(v[2 * p2 + 1]), false, false);
            u[p2] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
        }
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[2 * p2]), __float_as_uint(u[2 * p2 + 1]), false, false);
            w2[p2] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
        }
        if ((lane & 8) == 0) {
            const int rr = lane >> 4, idx = ((rr & 1) << 1) | (rr >> 1);
            red[tab][(wave * V4_PER_ROW + c4) * 8 + idx] = w2[0];
            red[tab][(wave * V4_PER_ROW + c4) * 8 + 4 + idx] = w2[1];
        }
        if (tid < V4_PER_ROW) *reinterpret_cast<float4*>(&redx0[tab][4 * c4]) = r.x0;
    };
    // lane f % 32 of every wave folds feature f of the slice (the four waves do the same work); a / s reach the staging threads through a wave-private table
    float4 av, sv;
    float fscale = 1.0f;                                 // the power-of-two scale of the f16 split
    float one = 1.0f;
    asm volatile("" : "+v"(one));
    auto fold = [&](const Slice& r, const int tab, const int k0) {
        const int f = lane & (BK - 1), fc = f >> 2, ft = f & 3;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            s1 += red[tab][(w * V4_PER_ROW + fc) * 8 + ft];
            s2 += red[tab][(w * V4_PER_ROW + fc) * 8 + 4 + ft];
        }
        const float inv_n = 1.0f / (float)N;
        const float m1 = s1 * inv_n;
        const float var = fmaxf(__builtin_fmaf(-m1, m1, s2 * inv_n), 0.f);      // biased variance (normalisation)
        const float mu = redx0[tab][f] + m1;
        const float ve = var + bo.eps;
        float rs = __builtin_amdgcn_rsqf(ve);
        rs = rs * __builtin_fmaf(-0.5f * ve * rs, rs, 1.5f);                     // one Newton step: v_rsq_f32 is ~1 ulp
        const bool fin = k0 + f < D;
        const float aa = fin ? (r.gsc + gdef) * rs : 0.f;                        // features past D: exact zeros whatever eps is
        const float ss = fin ? __builtin_fmaf(-mu, aa, r.bsc) : 0.f;
        float* tab_as = &fold_as[wave * 2 * BK];
        tab_as[f] = aa * fscale;                         // the staging threads get the map with the f16 scale folded in (a power of two: exact)
        tab_as[BK + f] = ss * fscale;
        const int o = (st_lane && fin) ? 4 * f : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mu), o_mean, o, k0 * 4, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rs), o_rstd, o, k0 * 4, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(aa), o_a, o, k0 * 4, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ss), o_s, o, k0 * 4, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(var * unb), o_var, o, k0 * 4, 0);
        av = *reinterpret_cast<const float4*>(&tab_as[4 * c4]);
        sv = *reinterpret_cast<const float4*>(&tab_as[BK + 4 * c4]);
    };
    auto lstore = [&](const Slice& r, const int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int row = (tid + 256 * i) / V4_PER_ROW;
            float y[4] = {__builtin_fmaf(av.x, r.rg[i].x, sv.x), __builtin_fmaf(av.y, r.rg[i].y, sv.y), __builtin_fmaf(av.z, r.rg[i].z, sv.z),
                          __builtin_fmaf(av.w, r.rg[i].w, sv.w)};              // S y
            f16x4 h, m;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (i == NLD - 1) y[t] = rowok[i] ? y[t] : 0.f;                // padded rows must stay 0 (their y would be s)
                h[t] = (_Float16)y[t];
                m[t] = (_Float16)__builtin_fmaf(y[t], one, -(float)h[t]);     // one v_fma_mix per element (`one` is opaque to the compiler: a plain y - h costs cvt + sub + cvt)
            }
            _Float16* dst = &zp[buf][row * SPLD + 4 * c4];
            *reinterpret_cast<f16x4*>(dst) = h;
            *reinterpret_cast<f16x4*>(dst + PLANE) = m;
        }
    };
    f32x4 acc[NT + 1];
#pragma unroll
    for (int i = 0; i <= NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto tiles = [&](auto rows, const _Float16* zs) {
        using R = decltype(rows);
        if constexpr (R::RA >= 0) {
            // the four wave branches stay opaque to each other: hipcc otherwise hoists the fragment reads they share above the branch and sinks the
            // `c += t` below the join -- every tile's fresh accumulator and eight fragments live at once (+68 VGPRs at NT = 7, spills at 3 workgroups per CU)
            asm volatile("" ::: "memory");
            sym_tiles_mfma_f16x2<NT, R::RA, R::RB, SPLD, PLANE>(acc, zs, r16, q);
            constexpr int NACC = R::RA + 1 + (R::RB >= 0 ? R::RB + 1 : 0);
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("" : "+v"(acc[i]));
        }
    };
    auto compute = [&](const int buf) {
        if (wave == 0) tiles(RowsOf<NT, 0>{}, zp[buf]);
        else if (wave == 1) tiles(RowsOf<NT, 1>{}, zp[buf]);
        else if (wave == 2) tiles(RowsOf<NT, 2>{}, zp[buf]);
        else tiles(RowsOf<NT, 3>{}, zp[buf]);
    };

    Slice r0, r1;
    const int nk = (D + BK - 1) / BK;
    gload(r0, 0);
    gload(r1, BK);
    {
        // the a-priori element bound of train-mode BatchNorm: max_k |beta_k| + |gamma_k| sqrt(N - 1)
        const float sq = sqrtf((float)(N > 1 ? N - 1 : 1));
        float bm = 0.f;
        for (int k = tid; k < D; k += 256) {
            const float g = bo.has_gamma ? G[k] : 1.0f, be = bo.has_beta ? Bt[k] : 0.0f;
            bm = fmaxf(bm, __builtin_fmaf(fabsf(g), sq, fabsf(be)));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) bm = fmaxf(bm, __shfl_xor(bm, o, DKT_WAVE));
        if (lane == 0) bmax_w[wave] = bm;
        if (tid == 0) bad = 0;
    }
    partial(r0, 0);
    __syncthreads();
    {
        const float bm = fmaxf(fmaxf(fmaxf(bmax_w[0], bmax_w[1]), fmaxf(bmax_w[2], bmax_w[3])), 1e-30f) * 1.01f;
        const int eb = (int)((__float_as_uint(bm) >> 23) & 0xffu) - 127;
        const int e = max(-60, min(60, 14 - eb));
        fscale = __uint_as_float((unsigned)(e + 127) << 23);
    }
    fold(r0, 0, 0);
    lstore(r0, 0);
    gload(r0, 2 * BK);
    partial(r1, 1);
    __syncthreads();
    // stage kt: image kt & 1 holds slice kt; `near` holds slice kt + 1 raw, its partial sums in table (kt + 1) & 1; `far` is in flight with slice kt + 2
    auto stage = [&](Slice& near, Slice& far, const int kt, const int par) {      // par = (kt + 1) & 1, a literal at both call sites
        fold(near, par, (kt + 1) * BK);
        lstore(near, par);
        gload(near, (kt + 3) * BK);
        compute(par ^ 1);
        partial(far, par ^ 1);
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        stage(r1, r0, kt, 1);
        stage(r0, r1, kt + 1, 0);
    }

    // ---- row norms from the diagonal of the scaled G' = S^2 Y Y^T, the a-posteriori check, then E_ij = G'_ij rho_i rho_j ----
    const float thr = (float)D * 16384.f;
    auto put_diag = [&](const f32x4& t, int row_blk) {
        if ((r16 >> 2) == q) {
            const int rr = r16 & 3;
            const float v = rr == 0 ? t[0] : rr == 1 ? t[1] : rr == 2 ? t[2] : t[3];
            if (16 * row_blk + r16 < N && !(v >= thr && v <= 3.0e38f)) bad = 1;
            rho[16 * row_blk + r16] = 1.0f / fmaxf(sqrtf(fmaxf(v, 0.f)), 1e-12f * fscale);      // rho_i / S
        }
    };
    auto diag_of = [&](auto w) {
        constexpr int W = decltype(w)::value;
        constexpr int RA = RowsOf<NT, W>::RA, RB = RowsOf<NT, W>::RB;
        if constexpr (RA >= 0) put_diag(acc[RA], RA);
        if constexpr (RB >= 0) put_diag(acc[RA + 1 + RB], RB);
    };
    if (wave == 0) diag_of(std::integral_constant<int, 0>{});
    else if (wave == 1) diag_of(std::integral_constant<int, 1>{});
    else if (wave == 2) diag_of(std::integral_constant<int, 2>{});
    else diag_of(std::integral_constant<int, 3>{});
    __syncthreads();
    const bool flagged = bad != 0;
    if (tid < N) rnorm[(size_t)b * N + tid] = (flagged && tid == 0) ? __uint_as_float(0x7fc00000u) : rho[tid] * fscale;
    if (flagged) return;                                 // E of this episode comes from the fix-up launch
    auto emit = [&](f32x4& t, const int rowblk, const int tj) {
        const float rj = rho[16 * tj + r16];
        const f32x4 ri = *reinterpret_cast<const f32x4*>(&rho[16 * rowblk + 4 * q]);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) t[reg] *= ri[reg] * rj;
    };
    auto finish = [&](auto w) {
        constexpr int W = decltype(w)::value;
        constexpr int RA = RowsOf<NT, W>::RA, RB = RowsOf<NT, W>::RB;
        if constexpr (RA >= 0) {
#pragma unroll
            for (int tj = 0; tj <= RA; ++tj) emit(acc[tj], RA, tj);
            sym_store_row<RA>(acc, Eb, N, r16, q);
        }
        if constexpr (RB >= 0) {
#pragma unroll
            for (int tj = 0; tj <= RB; ++tj) emit(acc[RA + 1 + tj], RB, tj);
            sym_store_row<RB>(acc + RA + 1, Eb, N, r16, q);
        }
    };
    if (wave == 0) finish(std::integral_constant<int, 0>{});
    else if (wave == 1) finish(std::integral_constant<int, 1>{});
    else if (wave == 2) finish(std::integral_constant<int, 2>{});
    else finish(std::integral_constant<int, 3>{});
}

// ---------------------------------------------------------------------------------------------
// Fused backward.  With A = g (W + W^T) (W = d obj / d E, g = upstream scale of the episode):
//   dZn = A Zn                                  the Gram backward, Zn re-created from X while it is staged
//   dY_i = rho_i (dZn_i - zn_i t_i),  t_i = zn_i . dZn_i = sum_j A_ij E_ij      (F.normalize backward; t needs no D-loop)
//   dbeta_d = sum_i dY_id,  dgamma_d = sum_i dY_id xh_id,  xh = (x - mean) rstd  (BatchNorm1d backward, batch statistics)
//   dX_id = a_d (dY_id - dbeta_d / N - xh_id dgamma_d / N)
// Structure of gram_bwd_ep_f16x2_kernel<NT, 1, 1> (round 3; bf16 x 3 before): the rows of Zn are unit-norm BY CONSTRUCTION here --
// rho comes from the forward -- so the staged operand takes the scaled 2-way f16 split (2^15, two planes, 3 MFMAs per product), and the A
// operand g (W + W^T) the per-row power-of-two scale of that kernel.  NT waves, wave w owns output rows [16w, 16w+16), its A fragments
// stay in registers; per 64-feature slab the MFMA result dZn is finished IN REGISTERS: the column sums over the N rows
// are reduced over the 4 row groups of a wave with shuffles and over the waves through a small LDS table that rides on
// the slab loop's barrier (image and table are double-buffered: ONE barrier per slab).  X is read once for staging and once (L2-hot) for
// the epilogue; dX is written once.
// Register budget: ~190 VGPRs = ONE workgroup of 7 waves per CU at N = 105.  Measured in round 3 (profiles/r03/v2_frontend_bwd_variants.log): forcing
// 128 VGPRs for two workgroups per CU spills 51 registers and is 1.7 x slower; moving the epilogue's operand loads behind the MFMA loop to save
// registers costs more than it gains at one workgroup per CU.
#ifndef DKT_FE_BWD_WPE
#define DKT_FE_BWD_WPE 2
#endif
template <int NT, bool TRAIN_BN>
__global__ __launch_bounds__(64 * NT, DKT_FE_BWD_WPE) void gram_bn_bwd_ep_kernel(const float* __restrict__ W, const float* __restrict__ Eg,
                                                                    const float* __restrict__ X, const float* __restrict__ Aa,
                                                                    const float* __restrict__ Ss, long ab_bstride,
                                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                    const float* __restrict__ rnorm, const float* __restrict__ ep_scale,
                                                                    float* __restrict__ dX, float* __restrict__ dgamma_part,
                                                                    float* __restrict__ dbeta_part, int N, int D) {
    constexpr int NP = 16 * NT;
    constexpr int NTH = 64 * NT;
    constexpr int BD = 64;
    constexpr int KS = (NP + 31) / 32;
    constexpr int KP = 32 * KS;
    constexpr int SU = (KP / 8) + ((KP / 8) % 4 == 2 ? 0 : (6 - (KP / 8) % 4) % 4);
    constexpr int RS = 8 * SU;
    constexpr int PLANE = BD * RS;
    constexpr int IMG = 2 * PLANE * 2;                   // bytes of the [d][j] image (two f16 planes)
    constexpr int STG = (2 * IMG > 2 * NP * NP * 4) ? 2 * IMG : 2 * NP * NP * 4;     // two image buffers; the region also stages W AND E (N x N fp32 each) together
    __shared__ __attribute__((aligned(16))) unsigned char smem[STG];
    __shared__ __attribute__((aligned(16))) float rl[NP], tl[NP], rowinv[NP];
    __shared__ __attribute__((aligned(16))) float cs[2][NT][16][8];          // per wave and lane column: the c1 / c2 sums of its 4 features; double-buffered like the image: ONE barrier per slab
    _Float16* zt = reinterpret_cast<_Float16*>(smem);
    float* wl = reinterpret_cast<float*>(smem);

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float g = ep_scale ? ep_scale[b] : 1.0f;
    const float inv_n = 1.0f / (float)N;
    const brsrc_t xr = mk_rsrc(X + (size_t)b * N * D, N * D * 4);
    const brsrc_t ar = mk_rsrc(Aa + (size_t)b * ab_bstride, D * 4);
    const brsrc_t sr = mk_rsrc(Ss + (size_t)b * ab_bstride, D * 4);
    const brsrc_t mr = mk_rsrc(TRAIN_BN ? mean + (size_t)b * D : Aa, D * 4);
    const brsrc_t rr_ = mk_rsrc(TRAIN_BN ? rstd + (size_t)b * D : Aa, D * 4);
    float* dXb = dX + (size_t)b * N * D;
    const brsrc_t dxr = mk_rsrc(dXb, N * D * 4);
    const brsrc_t dgr = mk_rsrc(TRAIN_BN ? dgamma_part + (size_t)b * D : dXb, TRAIN_BN ? D * 4 : 0);
    const brsrc_t dbr = mk_rsrc(TRAIN_BN ? dbeta_part + (size_t)b * D : dXb, TRAIN_BN ? D * 4 : 0);

    // staging task of this thread: rows 4 jg .. 4 jg + 3, features 4 d4 .. 4 d4 + 3 of the slab
    const int d4 = tid & 15, jg = tid >> 4;
    int voff[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) voff[rr] = (4 * jg + rr < N) ? ((4 * jg + rr) * D + 4 * d4) * 4 : OOB;
    float4 rg[4], sa, ss;
    auto gload = [&](int d0) {
        const bool in = d0 + 4 * d4 < D;
        sa = bload4(ar, in ? 16 * d4 : OOB, d0 * 4);
        ss = bload4(sr, in ? 16 * d4 : OOB, d0 * 4);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) rg[rr] = bload4(xr, in ? voff[rr] : OOB, d0 * 4);
    };
    const int nslab = (D + BD - 1) / BD;
    gload(0);                                            // flies while W and E are staged

    // ---- A fragments from W (as in gram_bwd_ep_bf16x3_kernel), then t_i = sum_k A_ik E_ik with E staged the same way ----
    f16x8 ah[KS], am[KS];
    const int nn = N * N;
    const int row = wave * 16 + r16;
    float rsinv;                                         // 1 / (row scale of A)
    {
        // W and E are staged TOGETHER (one global round trip and two barriers less than one after the other: with one workgroup per CU nothing
        // else covers the prologue, which the phase clocks put at a quarter of an episode)
        const float* Wb = W + (size_t)b * nn;
        const float* Eb = Eg + (size_t)b * nn;
        float* el = wl + NP * NP;
        DKT_LDS_STAGE_OLD_LOOP(for (int i = tid; i < nn; i += NTH) { wl[i] = Wb[i]; el[i] = Eb[i]; })
        {
            LdsStage<NTH, NT> wst, est;                 // all of W and E in flight at once (dkt_split.h)
            wst.load(Wb, nn, tid);
            est.load(Eb, nn, tid);
            wst.store(wl, nn, tid);
            est.store(el, nn, tid);
        }
        if (tid < NP) rl[tid] = (tid < N) ? rnorm[(size_t)b * N + tid] : 0.f;      // padded rows: rho = 0 -> dY = 0
        __syncthreads();
        float v[KS][8];
        float rmax = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 32 * ks + 8 * q + e;
                v[ks][e] = (row < N && k < N) ? g * (wl[row * N + k] + wl[k * N + row]) : 0.f;
                rmax = fmaxf(rmax, fabsf(v[ks][e]));
            }
        }
        rmax = fmaxf(rmax, __shfl_xor(rmax, 16, DKT_WAVE));
        rmax = fmaxf(rmax, __shfl_xor(rmax, 32, DKT_WAVE));
        // power-of-two row scale: row maximum -> [2^14, 2^15); clamped so that its inverse (times 2^-15) stays normal
        const int eb = (int)((__float_as_uint(rmax) >> 23) & 0xffu);
        const int sexp = min(268 - eb, 237);
        const float rscale = __uint_as_float((unsigned)sexp << 23);
        rsinv = __uint_as_float((unsigned)(254 - sexp) << 23);
        if (q == 0) rowinv[row] = __uint_as_float((unsigned)(254 - sexp - 15) << 23);      // undoes the row scale and the 2^15 of Zn
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
        float tp = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 32 * ks + 8 * q + e;
                const float aik = ((float)ah[ks][e] + (float)am[ks][e]) * rsinv;                // the A the MFMAs see (22 bits)
                if (row < N && k < N) tp = __builtin_fmaf(aik, el[row * N + k], tp);
            }
        }
        tp += __shfl_xor(tp, 16, DKT_WAVE);
        tp += __shfl_xor(tp, 32, DKT_WAVE);
        if (q == 0) tl[row] = tp;
        __syncthreads();
    }
    // this lane's output rows i = 16 wave + 4 q + reg: rho_i, t_i, the un-scaling of the row stay in registers for the whole episode
    // dY_i = rho_i (dZn_i - zn_i t_i) with dZn = un_i acc and zn = rho_i y:  dY = (rho un) acc - (rho^2 t) y -- two constants per row
    f32x4 ru4, r2t4;
    {
        const f32x4 rho4 = *reinterpret_cast<const f32x4*>(&rl[16 * wave + 4 * q]);
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(&tl[16 * wave + 4 * q]);
        const f32x4 un4 = *reinterpret_cast<const f32x4*>(&rowinv[16 * wave + 4 * q]);
        ru4 = rho4 * un4;
        r2t4 = rho4 * rho4 * t4;
    }
    const f32x4 rhoj = *reinterpret_cast<const f32x4*>(&rl[4 * jg]);      // rho_j of the staging task's rows
    __syncthreads();                                     // everyone is done with the staged E before the image is written
    if constexpr (KP > NP) {                             // columns j in [NP, KP) of the image are never staged: zero them once
        constexpr int PADV = (KP - NP) / 8;
        for (int i = tid; i < 2 * 2 * BD * PADV; i += NTH) {     // both buffers, both planes
            const int rowi = i / PADV, pc = i % PADV;
            *reinterpret_cast<float4*>(zt + (size_t)rowi * RS + NP + 8 * pc) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    auto lstore = [&](const int buf) {                   // zn = (a x + s) rho_j (|zn| <= 1), scaled 2-way f16 split, transposed 8-byte stores
        const float av[4] = {sa.x, sa.y, sa.z, sa.w}, sv[4] = {ss.x, ss.y, ss.z, ss.w};
        const float x[4][4] = {{rg[0].x, rg[1].x, rg[2].x, rg[3].x}, {rg[0].y, rg[1].y, rg[2].y, rg[3].y},
                               {rg[0].z, rg[1].z, rg[2].z, rg[3].z}, {rg[0].w, rg[1].w, rg[2].w, rg[3].w}};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float zn[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) zn[rr] = __builtin_fmaf(av[t], x[t][rr], sv[t]) * rhoj[rr];
            f16x4 h, m;
            split2h(make_float4(zn[0], zn[1], zn[2], zn[3]), DKT_F16_SCALE, h, m);
            _Float16* dst = zt + buf * 2 * PLANE + (16 * t + d4) * RS + 4 * jg;
            *reinterpret_cast<f16x4*>(dst) = h;
            *reinterpret_cast<f16x4*>(dst + PLANE) = m;
        }
    };

#ifdef DKT_FE_CLOCKS      // measurement build (tools/fe_bwd_clocks.py): shader clocks per phase of wave 0, summed over the slabs -> dgamma_part[b, 0..7]
    unsigned long long ck[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c0 = __builtin_amdgcn_s_memtime();
#define FCLK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long c1 = __builtin_amdgcn_s_memtime(); ck[i] += c1 - c0; c0 = c1; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FCLK(i) do { } while (0)
#endif
    __syncthreads();                                     // pad columns zeroed
    lstore(0);
    __syncthreads();
    FCLK(0);                                             // prologue (W, E staging, A fragments, first image)
    for (int sl = 0; sl < nslab; ++sl) {
        const int d0 = sl * BD, buf = sl & 1;
        // epilogue operands of this lane's 4 rows x 4 features (d = d0 + 4 r16 + t).  The lane's staging task IS that patch (jg = tid / 16 = 4 wave + q,
        // d4 = tid % 16 = r16): the x, a, s that were loaded to stage this slab's image are still in rg / sa / ss -- kept (round 5) instead of loaded a second
        // time (4 + 2 of the 9 sixteen-byte loads per thread and slab, a third of what the CU pulls through its L1).  Same-box A/B against the second load
        // (-DDKT_FE_BWD_RELOAD_X, tools/fe_bwd_lib_ab.py, profiles/r05/v6_fe_bwd_lib_ab.log; bitwise equal): N = 105 / D = 1600 0.771 -> 0.724 ms per 2048
        // episodes, 3.06 -> 2.83 ms per 8192, N = 128: 0.957 -> 0.865 ms; at NT <= 6 the second load stays -- there the shorter live ranges of the reuse
        // form let a second workgroup onto the CU (162 instead of 174 VGPRs) and that measured 2 - 3 % SLOWER (N = 85 / D = 512: 0.253 -> 0.260 ms).
        const bool din = d0 + 4 * r16 < D;
        float4 xe[4], ea, es;
#ifdef DKT_FE_BWD_RELOAD_X
        constexpr bool REUSE = false;
#else
        constexpr bool REUSE = NT >= 7;
#endif
        // (this slab's own operands are issued BEFORE the next slab's staging loads: the counter is in order, and a wait for mean / rstd issued behind the
        // prefetch -- the round-3 order -- was a `vmcnt(0)` that drained the prefetch in front of every epilogue)
        float4 em = make_float4(0.f, 0.f, 0.f, 0.f), er = em;
        if constexpr (TRAIN_BN) {
            em = bload4(mr, din ? 16 * r16 : OOB, d0 * 4);
            er = bload4(rr_, din ? 16 * r16 : OOB, d0 * 4);
        }
        if constexpr (REUSE) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) xe[reg] = rg[reg];
            ea = sa;
            es = ss;
        } else {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = 16 * wave + 4 * q + reg;
                xe[reg] = bload4(xr, (din && i < N) ? (i * D + 4 * r16) * 4 : OOB, d0 * 4);
            }
            ea = bload4(ar, din ? 16 * r16 : OOB, d0 * 4);
            es = bload4(sr, din ? 16 * r16 : OOB, d0 * 4);
        }
        gload(d0 + BD);                                  // unconditional (past D: every offset out of range, zeros): behind a branch the compiler cannot count these six loads
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const _Float16* base = zt + buf * 2 * PLANE + r16 * RS + 8 * q;
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
        FCLK(1);                                         // loads issued + MFMA loop (LDS fragment reads)
#ifdef DKT_FE_CLOCKS
        __builtin_amdgcn_s_waitcnt(0x0070 | 0x0f00 | 0xc000);      // vmcnt(0) expcnt(7) lgkmcnt... : the epilogue operands (and the prefetch) have arrived
        FCLK(2);
#endif
        // dY and the normalised inputs xh in packed fp32 (pairs of adjacent features: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 halve the
        // VALU instructions of what the phase clocks show to be the heaviest part of a slab)
        const f32x2 a2[2] = {{ea.x, ea.y}, {ea.z, ea.w}}, s2[2] = {{es.x, es.y}, {es.z, es.w}};
        const f32x2 er2[2] = {{er.x, er.y}, {er.z, er.w}};
        const f32x2 emer2[2] = {{-em.x * er.x, -em.y * er.y}, {-em.z * er.z, -em.w * er.w}};       // xh = x rstd - mean rstd
        f32x2 dy2[4][2], xh2[4][2];                     // [reg][feature pair]
        f32x2 c1p[2] = {{0.f, 0.f}, {0.f, 0.f}}, c2p[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const f32x2 x2[2] = {{xe[reg].x, xe[reg].y}, {xe[reg].z, xe[reg].w}};
            const f32x2 ru = {ru4[reg], ru4[reg]}, r2t = {r2t4[reg], r2t4[reg]};
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const f32x2 y = a2[p2] * x2[p2] + s2[p2];
                const f32x2 ac = {acc[2 * p2][reg], acc[2 * p2 + 1][reg]};
                const f32x2 dy = ac * ru - r2t * y;
                dy2[reg][p2] = dy;
                if constexpr (TRAIN_BN) {
                    const f32x2 xh = x2[p2] * er2[p2] + emer2[p2];
                    xh2[reg][p2] = xh;
                    c1p[p2] += dy;
                    c2p[p2] += dy * xh;
                }
            }
        }
        if constexpr (TRAIN_BN) {
            // over the 4 row groups q of the wave: reduce-scatter with v_permlane32_swap / v_permlane16_swap on PAIRS of values (one swap + one add
            // halves the live values); row group q ends up with the sums number {0, 2, 1, 3}[q] (of c1) and 4 + that (of c2) of its column r16
            const float v[8] = {c1p[0].x, c1p[0].y, c1p[1].x, c1p[1].y, c2p[0].x, c2p[0].y, c2p[1].x, c2p[1].y};
            float u[4], w2[2];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * h]), __float_as_uint(v[2 * h + 1]), false, false);
                u[h] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(u[2 * h]), __float_as_uint(u[2 * h + 1]), false, false);
                w2[h] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            const int idx = ((q & 1) << 1) | (q >> 1);
            float* dst = &cs[buf][wave][r16][0];
            dst[idx] = w2[0];
            dst[4 + idx] = w2[1];
        }
        FCLK(3);                                         // first half of the epilogue
        lstore(buf ^ 1);                                 // the other image buffer: last read before the previous barrier (unconditional: behind the last slab it stages zeros)
        FCLK(4);                                         // split + LDS stores of the next image
        __syncthreads();                                 // next image staged; column partials of every wave published
        FCLK(5);                                         // barrier wait
        f32x2 m1p[2] = {{0.f, 0.f}, {0.f, 0.f}}, m2p[2] = {{0.f, 0.f}, {0.f, 0.f}};
        if constexpr (TRAIN_BN) {
#pragma unroll
            for (int w = 0; w < NT; ++w) {               // fixed order over the waves: deterministic
                const f32x4 p1 = *reinterpret_cast<const f32x4*>(&cs[buf][w][r16][0]);
                const f32x4 p2 = *reinterpret_cast<const f32x4*>(&cs[buf][w][r16][4]);
                m1p[0] += (f32x2){p1[0], p1[1]}; m1p[1] += (f32x2){p1[2], p1[3]};
                m2p[0] += (f32x2){p2[0], p2[1]}; m2p[1] += (f32x2){p2[2], p2[3]};
            }
            // (branch-free stores, round 5: lanes with nothing to store get an out-of-range offset -- behind a branch the compiler counts no store as
            // outstanding, and its `vmcnt(0)` in front of the next slab's operands drained these stores at the top of every trip; soffset stays 0)
            const int po = (wave == 0 && q == 0 && din) ? (d0 + 4 * r16) * 4 : OOB;
            fe_store4(dbr, po, m1p[0].x, m1p[0].y, m1p[1].x, m1p[1].y);
            fe_store4(dgr, po, m2p[0].x, m2p[0].y, m2p[1].x, m2p[1].y);
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                m1p[p2] *= inv_n;
                m2p[p2] *= inv_n;
            }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = 16 * wave + 4 * q + reg;
            f32x2 o2[2];
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                f32x2 vv = dy2[reg][p2];
                if constexpr (TRAIN_BN) vv = vv - m1p[p2] - xh2[reg][p2] * m2p[p2];
                o2[p2] = a2[p2] * vv;
            }
            fe_store4(dxr, (i < N && din) ? (i * D + d0 + 4 * r16) * 4 : OOB, o2[0].x, o2[0].y, o2[1].x, o2[1].y);
        }
        FCLK(6);                                         // second half of the epilogue + dX stores
        // no second barrier: cs[buf] and image[buf] are written again two slabs on, i.e. behind the next slab's barrier
    }
#ifdef DKT_FE_CLOCKS
    if (tid == 0 && dgamma_part) {
#pragma unroll
        for (int i = 0; i < 7; ++i) dgamma_part[(size_t)b * D + i] = (float)ck[i];
    }
#endif
#undef FCLK
}

static int g_stage_synced = 0;                          // twins library: DKT_LDS_STAGE_OLD reaches the device at the first launch and after dkt_reload_env()
void lds_stage_env_sync_once() {
    if (!g_stage_synced) { lds_stage_env_sync(); g_stage_synced = 1; }
}

template <int NT>
void launch_gram_bn_bwd(const float* W, const float* E, const float* X, const float* a, const float* s, long abs, const float* mean,
                        const float* rstd, const float* rnorm, const float* sc, float* dX, float* dg, float* db, int B, int N, int D,
                        bool train_bn, hipStream_t st) {
    lds_stage_env_sync_once();
    if (train_bn) hipLaunchKernelGGL((gram_bn_bwd_ep_kernel<NT, true>), dim3(B), dim3(64 * NT), 0, st, W, E, X, a, s, abs, mean, rstd, rnorm, sc, dX, dg, db, N, D);
    else hipLaunchKernelGGL((gram_bn_bwd_ep_kernel<NT, false>), dim3(B), dim3(64 * NT), 0, st, W, E, X, a, s, abs, mean, rstd, rnorm, sc, dX, dg, db, N, D);
}

// DKT_GRAM_BN_F16 (twins library only; default 1): 0 = the 3-way bf16 split in the train-mode fused forward at every N (the round-3 kernel: A/B, twin test)
static int g_bn_f16 = -1;
bool bn_train_f16() {
    if (g_bn_f16 < 0) { const char* v = dkt_variant_env("DKT_GRAM_BN_F16"); g_bn_f16 = (v && v[0] == '0') ? 0 : 1; }
    return g_bn_f16 != 0;
}

template <int NT>
void launch_gram_bn(const float* X, const float* A, const float* S, long abs, float* E, float* rnorm, int B, int N, int D, hipStream_t st,
                    const BnTrainOut* bo) {
    if (bo && bo->epi != 0) {
        if (bo->epi == 2) hipLaunchKernelGGL((gram_bn_sym_ep_kernel<NT, false, 2>), dim3(B), dim3(256), 0, st, X, X, X, 0L, E, nullptr, N, D, *bo, 0);       // RBF
        else hipLaunchKernelGGL((gram_bn_sym_ep_kernel<NT, false, 1>), dim3(B), dim3(256), 0, st, X, X, X, 0L, E, nullptr, N, D, *bo, 0);                     // SQDIST
    } else if (bo) {
        if constexpr (NT >= 3) {
            if (bn_train_f16()) {                        // f16 split under the a-priori bound, then the fix-up pass over the episodes it flagged
                hipLaunchKernelGGL((gram_bn_train_f16_kernel<NT>), dim3(B), dim3(256), dkt_lds_pad("DKT_PAD_FE_FWD"), st, X, A, S, E, rnorm, N, D, *bo);
                hipLaunchKernelGGL((gram_bn_sym_ep_kernel<NT, true>), dim3(B), dim3(256), dkt_lds_pad("DKT_PAD_FE_FWD"), st, X, A, S, 0L, E, rnorm, N, D, *bo, 1);
                return;
            }
        }
        hipLaunchKernelGGL((gram_bn_sym_ep_kernel<NT, true>), dim3(B), dim3(256), dkt_lds_pad("DKT_PAD_FE_FWD"), st, X, A, S, 0L, E, rnorm, N, D, *bo, 0);
    } else hipLaunchKernelGGL((gram_bn_sym_ep_kernel<NT, false>), dim3(B), dim3(256), 0, st, X, A, S, abs, E, rnorm, N, D, BnTrainOut{}, 0);
}

int gram_bn_dispatch(const float* X, const float* a, const float* s, long abs, float* E, float* rnorm, int B, int N, int D, hipStream_t st,
                     const BnTrainOut* bo) {
    switch ((N + 15) / 16) {
        case 1: launch_gram_bn<1>(X, a, s, abs, E, rnorm, B, N, D, st, bo); break;
        case 2: launch_gram_bn<2>(X, a, s, abs, E, rnorm, B, N, D, st, bo); break;
        case 3: launch_gram_bn<3>(X, a, s, abs, E, rnorm, B, N, D, st, bo); break;
        case 4: launch_gram_bn<4>(X, a, s, abs, E, rnorm, B, N, D, st, bo); break;
        case 5: launch_gram_bn<5>(X, a, s, abs, E, rnorm, B, N, D, st, bo); break;
        case 6: launch_gram_bn<6>(X, a, s, abs, E, rnorm, B, N, D, st, bo); break;
        case 7: launch_gram_bn<7>(X, a, s, abs, E, rnorm, B, N, D, st, bo); break;
        default: launch_gram_bn<8>(X, a, s, abs, E, rnorm, B, N, D, st, bo); break;
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

}  // namespace

// Episode-resident squared-distance / RBF build of dkt_gram_f32 (symmetric, 32 < N <= 128, D % 4 == 0, 16-byte aligned Z, a batch that
// fills the GPU); returns false when it does not apply (the generic 64 x 64-tile kernel then runs).
static int g_dist_ep_minb = -1, g_dist_ep_on = -1;           // DKT_GRAM_EP_MINB / DKT_GRAM_DIST_EP, read at the first call and at dkt_reload_env()
void dkt_frontend_reload_env() { g_dist_ep_minb = -1; g_dist_ep_on = -1; g_bn_f16 = -1; g_stage_synced = 0; }
bool dkt_gram_dist_ep_launch(const float* Z, float* E, int B, int N, int D, int kind, const float* lengthscale, hipStream_t st) {
    if (g_dist_ep_minb < 0) { const char* v = getenv("DKT_GRAM_EP_MINB"); g_dist_ep_minb = v ? atoi(v) : 64; }
    if (g_dist_ep_on < 0) { const char* v = dkt_variant_env("DKT_GRAM_DIST_EP"); g_dist_ep_on = (v && v[0] == '0') ? 0 : 1; }
    const int minb = g_dist_ep_minb;
    const bool on = g_dist_ep_on != 0;
    if (!on || N <= 32 || N > 128 || (D & 3) || ((uintptr_t)Z & 15) || B < minb || !lengthscale) return false;
    if (kind != DKT_KERNEL_RBF && kind != DKT_KERNEL_SQDIST) return false;
    BnTrainOut bo{};
    bo.lengthscale = lengthscale;
    bo.epi = (kind == DKT_KERNEL_RBF) ? 2 : 1;
    gram_bn_dispatch(Z, Z, Z, 0, E, nullptr, B, N, D, st, &bo);
    return true;
}

extern "C" int dkt_bn_stats_f32(const float* X, const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                                float* a, float* s, float* var_unbiased, int B, int N, int D, void* stream) {
    if (!X || !mean || !rstd || !a || !s || B <= 0 || N <= 0 || D <= 0) return DKT_ERR_BAD_ARG;
    if ((D & 3) || ((uintptr_t)X & 15)) return DKT_ERR_BAD_ARG;
    if (B > 65535) return DKT_ERR_TOO_LARGE;
    dim3 grid((D + 255) / 256, B);
    hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(1024), 0, (hipStream_t)stream, X, gamma, beta, eps, mean, rstd, a, s, var_unbiased, N, D);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_gram_bn_f32(const float* X, const float* a, const float* s, long ab_bstride, float* E, float* rnorm,
                               int B, int N, int D, void* stream) {
    if (!X || !a || !s || !E || !rnorm || B <= 0 || N <= 0 || D <= 0) return DKT_ERR_BAD_ARG;
    if ((D & 3) || ((uintptr_t)X & 15) || ((uintptr_t)a & 15) || ((uintptr_t)s & 15) || (ab_bstride & 3)) return DKT_ERR_BAD_ARG;
    if (N > 128) return DKT_ERR_TOO_LARGE;
    return gram_bn_dispatch(X, a, s, ab_bstride, E, rnorm, B, N, D, (hipStream_t)stream, nullptr);
}

extern "C" int dkt_gram_bn_train_f32(const float* X, const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                                     float* a, float* s, float* var_unbiased, float* E, float* rnorm, int B, int N, int D, void* stream) {
    if (!X || !mean || !rstd || !a || !s || !E || !rnorm || B <= 0 || N <= 0 || D <= 0) return DKT_ERR_BAD_ARG;
    if ((D & 3) || ((uintptr_t)X & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) return DKT_ERR_BAD_ARG;
    if (((uintptr_t)mean & 15) || ((uintptr_t)rstd & 15) || ((uintptr_t)a & 15) || ((uintptr_t)s & 15) || ((uintptr_t)var_unbiased & 15)) return DKT_ERR_BAD_ARG;
    if (N > 128) return DKT_ERR_TOO_LARGE;
    BnTrainOut bo{};
    bo.mean = mean; bo.rstd = rstd; bo.a = a; bo.s = s; bo.var_unbiased = var_unbiased;
    bo.eps = eps; bo.has_gamma = gamma != nullptr; bo.has_beta = beta != nullptr;
    // absent gamma / beta: any valid pointer keeps the descriptor legal, the values are ignored (has_* = 0)
    return gram_bn_dispatch(X, gamma ? gamma : X, beta ? beta : X, 0, E, rnorm, B, N, D, (hipStream_t)stream, &bo);
}

extern "C" int dkt_gram_bn_bwd_f32(const float* W, const float* E, const float* X, const float* a, const float* s, long ab_bstride,
                                   const float* mean, const float* rstd, const float* rnorm, const float* ep_scale, float* dX,
                                   float* dgamma_part, float* dbeta_part, int B, int N, int D, void* stream) {
    if (!W || !E || !X || !a || !s || !rnorm || !dX || B <= 0 || N <= 0 || D <= 0) return DKT_ERR_BAD_ARG;
    const bool train_bn = mean != nullptr;
    if (train_bn && (!rstd || !dgamma_part || !dbeta_part)) return DKT_ERR_BAD_ARG;
    if ((D & 3) || ((uintptr_t)X & 15) || ((uintptr_t)dX & 15) || ((uintptr_t)a & 15) || ((uintptr_t)s & 15) || (ab_bstride & 3)) return DKT_ERR_BAD_ARG;
    if (N > 128) return DKT_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
    switch ((N + 15) / 16) {
        case 1: launch_gram_bn_bwd<1>(W, E, X, a, s, ab_bstride, mean, rstd, rnorm, ep_scale, dX, dgamma_part, dbeta_part, B, N, D, train_bn, st); break;
        case 2: launch_gram_bn_bwd<2>(W, E, X, a, s, ab_bstride, mean, rstd, rnorm, ep_scale, dX, dgamma_part, dbeta_part, B, N, D, train_bn, st); break;
        case 3: launch_gram_bn_bwd<3>(W, E, X, a, s, ab_bstride, mean, rstd, rnorm, ep_scale, dX, dgamma_part, dbeta_part, B, N, D, train_bn, st); break;
        case 4: launch_gram_bn_bwd<4>(W, E, X, a, s, ab_bstride, mean, rstd, rnorm, ep_scale, dX, dgamma_part, dbeta_part, B, N, D, train_bn, st); break;
        case 5: launch_gram_bn_bwd<5>(W, E, X, a, s, ab_bstride, mean, rstd, rnorm, ep_scale, dX, dgamma_part, dbeta_part, B, N, D, train_bn, st); break;
        case 6: launch_gram_bn_bwd<6>(W, E, X, a, s, ab_bstride, mean, rstd, rnorm, ep_scale, dX, dgamma_part, dbeta_part, B, N, D, train_bn, st); break;
        case 7: launch_gram_bn_bwd<7>(W, E, X, a, s, ab_bstride, mean, rstd, rnorm, ep_scale, dX, dgamma_part, dbeta_part, B, N, D, train_bn, st); break;
        default: launch_gram_bn_bwd<8>(W, E, X, a, s, ab_bstride, mean, rstd, rnorm, ep_scale, dX, dgamma_part, dbeta_part, B, N, D, train_bn, st); break;
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
