// NOT PART OF ANY LIBRARY -- kept for the record (tools/experiments/README.md).  Built in round 5, parity-green at the first run, measured at the
// speed of the kernel it was meant to replace (0.3365 vs 0.3337 ms per 8192 cfg1 episodes), ablated (-DLS_EXP_NOSWEEP 0.251 ms, -DLS_EXP_NOP3 0.290 ms), not shipped.
//
// dkt_lowrank_solve.hip -- the 64 x 64 models of the feature-space episode (dkt_lowrank.hip): per episode the C matrices K'_c = sv_c A + noise_c I,
// their jittered factorisation, log det, t_c = K'_c^-1 p_c, tr K'_c^-1 and W' = 0.5 sum_c cw_c sv_c (t_c t_c^T - K'_c^-1).
//
// dkt_mll_f32 on (A, P) with N' = 64 is correct, but its augmented row makes N' + 1 = 65 a FIVE-tile problem (125 instead of 64 tile-product units, five instead
// of four diagonal sweeps).  Here the same algorithm (dkt_mll_mfma.hip: upper blocked K'/kappa = R^T R on v_mfma_f32_16x16x4_f32 with every tile in the
// accumulator layout, diagonal tiles by the DPP sweep of dkt_mfma_tiles.h, M = R^-T, K'^-1 = M^T M / kappa) runs on exactly 4 x 4 tiles, and the right-hand
// side rides along as a BORDER COLUMN instead of an augmented row:
//     S_k4 = -p_k / sqrt(kappa)  ->  panel R_k4 = (-V_kk)^T S_k4 = w_k,  trailing S_i4 += R_ki^T R_k4      (w = R^-T p: forward substitution inside the factorisation)
//     quad = |w|^2 = p^T K'^-1 p,     t_i = (1 / sqrt(kappa)) sum_{k >= i} M_ki^T w_k                         (back substitution with the tiles of M as they are)
// Exact fp32 products.  One wave per episode, the classes in sequence, W' accumulated over the classes in registers and stored once, mirrored.  Outputs in the
// conventions of dkt_mll_f32: T = t, Wd = W', logp_d = -0.5 quad - 0.5 log det K' - 32 log 2 pi, dnoise_d = 0.5 (t.t - tr K'^-1), jitter_used, info.
#include "dkt_mfma_tiles.h"

namespace {

using namespace dkt_mfma;

constexpr int LS_DP = 64;
constexpr int LS_NT = LS_DP / 16;              // 4

// four / two independent X^T Y chains advanced together (a dependent accumulate waits for the MFMA in front of it)
#define LS_XTY4(c0, x0, y0, c1, x1, y1, c2, x2, y2, c3, x3, y3)                                 \
    do {                                                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                      \
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((x0)[q_], (y0)[q_], c0, 0, 0, 0);         \
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((x1)[q_], (y1)[q_], c1, 0, 0, 0);         \
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32((x2)[q_], (y2)[q_], c2, 0, 0, 0);         \
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32((x3)[q_], (y3)[q_], c3, 0, 0, 0);         \
        }                                                                                       \
    } while (0)
#define LS_XTY2(c0, x0, y0, c1, x1, y1)                                                         \
    do {                                                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                      \
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32((x0)[q_], (y0)[q_], c0, 0, 0, 0);         \
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32((x1)[q_], (y1)[q_], c1, 0, 0, 0);         \
        }                                                                                       \
    } while (0)

__global__ __launch_bounds__(64) void lowrank_solve_kernel(const float* __restrict__ A, const float* __restrict__ P, const float* __restrict__ sv,
                                                           const float* __restrict__ noise, const float* __restrict__ cls_weight, const float jitter0,
                                                           const int max_tries, float* __restrict__ T, float* __restrict__ Wd, float* __restrict__ logp_d,
                                                           float* __restrict__ dnoise_d, float* __restrict__ jitter_used, int32_t* __restrict__ info, const int C) {
    __shared__ __attribute__((aligned(16))) float tl[LS_DP];      // t of the current class: the rank-one term reads it by row and by column
    const int b = blockIdx.x, lane = threadIdx.x, c16 = lane & 15, g4 = (lane >> 2) & 12;
    Lane ln;
    ln.lane = lane; ln.g = lane >> 4; ln.c = c16;
    ln.g0 = ln.g == 0; ln.g1 = ln.g == 1; ln.g2 = ln.g == 2;
    const float qnan = __int_as_float(0x7fc00000);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 negI;
#pragma unroll
    for (int q = 0; q < 4; ++q) negI[q] = (g4 + q == c16) ? -1.0f : 0.0f;
    const brsrc Ar = mk_rsrc(A + (size_t)b * LS_DP * LS_DP, LS_DP * LS_DP * 4);
    // tile (i, j), i <= j, of the symmetric A in the accumulator layout: element [4g + q][c] = A[16 j + c][16 i + 4 g + q] -- one 16-byte load per lane
    auto load_a = [&](const int i, const int j) { return bload4(Ar, ((16 * j + c16) * LS_DP + 16 * i + g4) * 4, 0); };
    float amax = 0.f;                             // max_i A_ii (for kappa)
#pragma unroll
    for (int k = 0; k < LS_NT; ++k) {
        const f32x4 e = load_a(k, k);
#pragma unroll
        for (int q = 0; q < 4; ++q) amax = fmaxf(amax, (g4 + q == c16) ? e[q] : 0.f);
    }
    amax = wave_reduce_dpp<true>(amax);

    f32x4 wacc[10];                               // W' tiles (i <= j), index j (j + 1) / 2 + i
#pragma unroll
    for (int n = 0; n < 10; ++n) wacc[n] = zero4;
    bool poison = false;

    for (int c = 0; c < C; ++c) {
        const float svc = sv[c], nzc = noise[c], cwc = cls_weight ? cls_weight[c] : 1.0f;
        const size_t bc = (size_t)b * C + c;
        const brsrc Pr = mk_rsrc(P + bc * LS_DP, LS_DP * 4);
        f32x4 t[LS_NT][LS_NT], md[LS_NT], w4[LS_NT];
        int fail_at = 0, msc = 0;
        float jit = 0.f, lsum = 0.f;
        for (int attempt = 0; attempt <= max_tries; ++attempt) {
            jit = 0.f;
            if (attempt > 0) {
                jit = jitter0;
                for (int i = 1; i < attempt; ++i) jit *= 10.f;
            }
            // kappa = 4^msc >= max_i K'_ii: every pivot of K' / kappa is <= 1 (the sweep divides by them)
            int ex;
            (void)frexpf(fmaf(svc, amax, nzc + jit), &ex);
            msc = max(0, (ex + 1) >> 1);
            const float ikap = ldexpf(1.0f, -2 * msc), isq = ldexpf(1.0f, -msc);
            const float nsv = -svc * ikap, dg = -(nzc + jit) * ikap;
            // ---- S = -K' / kappa (upper tiles) and the border column -p / sqrt(kappa) (column 0 of four tiles) ----
#pragma unroll
            for (int j = 0; j < LS_NT; ++j) {
#pragma unroll
                for (int i = 0; i <= j; ++i) {
                    const f32x4 e = load_a(i, j);
                    f32x4 s;
#pragma unroll
                    for (int q = 0; q < 4; ++q) s[q] = (i == j && g4 + q == c16) ? fmaf(nsv, e[q], dg) : nsv * e[q];
                    t[i][j] = s;
                }
            }
#pragma unroll
            for (int i = 0; i < LS_NT; ++i) {
                const f32x4 pv = bload4(Pr, (c16 == 0) ? (16 * i + g4) * 4 : OOB, 0);          // lanes c != 0: an out-of-range offset, 0
                w4[i] = pv * (-isq);
            }
            // ---- phase 1: factorisation, the border column carried along ----
            fail_at = 0;
            lsum = 0.f;
#pragma unroll
            for (int k = 0; k < LS_NT; ++k) {
                float x[16], dv;
#ifdef LS_EXP_NOSWEEP
                dv = 1.0f;
                const f32x4 M = t[k][k];
#else
                sweep_begin(t[k][k], x, dv);
                sweep_plain<0, false>(x, dv, ln, 0);
                const f32x4 M = sweep_end(x, ln);
#endif
                const unsigned long long badm = __ballot(!(dv > 0.f)) & 0xffffull;               // lanes 0 .. 15: pivot c of the tile
                const int first = (int)__builtin_ctzll(badm | 0x10000ull);
                fail_at = (fail_at == 0 && badm != 0) ? 16 * k + first + 1 : fail_at;
                lsum += ln.g0 ? __builtin_amdgcn_logf(dv) : 0.f;                                  // log2 of the pivots
                md[k] = M;
                const f32x4 nV = xty0(M, negI);                                                   // M^T (-I) = -V_kk
                // panel: R_kj = (-V_kk)^T S_kj, j > k, and the border tile; trailing: S_ij += R_ki^T R_kj
                if (k == 0) {
                    f32x4 r1 = zero4, r2 = zero4, r3 = zero4, r4 = zero4;
                    LS_XTY4(r1, nV, t[0][1], r2, nV, t[0][2], r3, nV, t[0][3], r4, nV, w4[0]);
                    t[0][1] = r1; t[0][2] = r2; t[0][3] = r3; w4[0] = r4;
                    LS_XTY4(t[1][1], t[0][1], t[0][1], t[1][2], t[0][1], t[0][2], t[1][3], t[0][1], t[0][3], w4[1], t[0][1], w4[0]);
                    LS_XTY4(t[2][2], t[0][2], t[0][2], t[2][3], t[0][2], t[0][3], w4[2], t[0][2], w4[0], t[3][3], t[0][3], t[0][3]);
                    w4[3] = xty(t[0][3], w4[0], w4[3]);
                } else if (k == 1) {
                    f32x4 r2 = zero4, r3 = zero4, r4 = zero4;
                    LS_XTY2(r2, nV, t[1][2], r3, nV, t[1][3]);
                    r4 = xty0(nV, w4[1]);
                    t[1][2] = r2; t[1][3] = r3; w4[1] = r4;
                    LS_XTY4(t[2][2], t[1][2], t[1][2], t[2][3], t[1][2], t[1][3], w4[2], t[1][2], w4[1], t[3][3], t[1][3], t[1][3]);
                    w4[3] = xty(t[1][3], w4[1], w4[3]);
                } else if (k == 2) {
                    f32x4 r3 = zero4, r4 = zero4;
                    LS_XTY2(r3, nV, t[2][3], r4, nV, w4[2]);
                    t[2][3] = r3; w4[2] = r4;
                    LS_XTY2(t[3][3], t[2][3], t[2][3], w4[3], t[2][3], w4[2]);
                } else {
                    w4[3] = xty0(nV, w4[3]);
                }
            }
            if (fail_at == 0) break;
        }
        const bool ok = fail_at == 0;
        // quad = |w|^2: column 0 of the four border tiles (the other lanes hold exact zeros)
        float quad = 0.f;
#pragma unroll
        for (int i = 0; i < LS_NT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) quad = fmaf(w4[i][q], w4[i][q], quad);
        quad = wave_reduce_dpp<false>(quad);
        lsum = wave_reduce_dpp<false>(lsum);
        // ---- phase 2: M = R^-T; M_ji (j > i) overwrites slot (i, j) ----
#ifndef LS_EXP_NOP2
#pragma unroll
        for (int j = 1; j < LS_NT; ++j) {
            const f32x4 nV = xty0(md[j], negI);
#pragma unroll
            for (int i = 0; i < j; ++i) {
                f32x4 Q = xty0(t[i][j], md[i]);                                                // k = i: R_ij^T M_ii
#pragma unroll
                for (int k = i + 1; k < j; ++k) Q = xty(t[k][j], t[i][k], Q);                  // R_kj^T M_ki
                t[i][j] = xty0(nV, Q);
            }
        }
#endif
        // ---- t = M^T w / sqrt(kappa): t_i = sum_{k >= i} M_ki^T w_k (M_ii = md[i], M_ki = slot (i, k)) ----
        f32x4 tv[LS_NT];
        {
            f32x4 a0 = zero4, a1 = zero4, a2 = zero4, a3 = zero4;
            LS_XTY4(a0, md[0], w4[0], a1, md[1], w4[1], a2, md[2], w4[2], a3, md[3], w4[3]);
            LS_XTY4(a0, t[0][1], w4[1], a1, t[1][2], w4[2], a2, t[2][3], w4[3], a0, t[0][2], w4[2]);
            LS_XTY2(a1, t[1][3], w4[3], a0, t[0][3], w4[3]);
            const float isq = ldexpf(1.0f, -msc);
            tv[0] = a0 * isq; tv[1] = a1 * isq; tv[2] = a2 * isq; tv[3] = a3 * isq;
        }
        // t -> T[b, c, :] and LDS (lanes c == 0 hold t[16 i + 4 g + q]); |t|^2
        const brsrc Tr = mk_rsrc(T + bc * LS_DP, LS_DP * 4);
        float tt = 0.f;
#pragma unroll
        for (int i = 0; i < LS_NT; ++i) {
            f32x4 o = tv[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tt = (c16 == 0) ? fmaf(o[q], o[q], tt) : tt;
                o[q] = ok ? o[q] : qnan;
            }
            bstore4(Tr, o, (c16 == 0) ? (16 * i + g4) * 4 : OOB, 0);
            if (c16 == 0) *reinterpret_cast<f32x4*>(&tl[16 * i + g4]) = tv[i];
        }
        tt = wave_reduce_dpp<false>(tt);
        __builtin_amdgcn_s_waitcnt(0xc07f);                                                    // lgkmcnt(0) (one wave per workgroup: LDS operations are in order)
        // ---- phase 3: P = M^T M = kappa K'^-1, accumulated into W' with the class coefficient; tr K'^-1 ----
        const float ikap = ldexpf(1.0f, -2 * msc);
        const float coefK = ok ? -0.5f * cwc * svc * ikap : qnan, coefA = 0.5f * cwc * svc;
        poison = poison || !ok;
        float trk = 0.f;
#pragma unroll
        for (int j = 0; j < LS_NT; ++j) {
#pragma unroll
            for (int i = 0; i <= j; ++i) {
                f32x4 acc = zero4;
#ifndef LS_EXP_NOP3
#pragma unroll
                for (int k = j; k < LS_NT; ++k) {
                    const f32x4 Bm = (k == j) ? md[j] : t[j][k];
                    const f32x4 A0 = (k == i) ? md[i] : t[i][k];
                    acc = xty(A0, Bm, acc);
                }
#else
                acc = t[i][j];
#endif
                const f32x4 tr = *reinterpret_cast<const f32x4*>(&tl[16 * i + g4]);           // t[16 i + 4 g + q]
                const float tc = tl[16 * j + c16] * coefA;
                const int n = j * (j + 1) / 2 + i;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    wacc[n][q] += fmaf(tr[q], tc, acc[q] * coefK);
                    if (i == j) trk += (g4 + q == c16) ? acc[q] : 0.f;
                }
            }
        }
        trk = wave_reduce_dpp<false>(trk) * ikap;
        if (lane == 0) {
            const float logdet = 0.69314718055994531f * (lsum + (float)(2 * msc * LS_DP));   // ln det K' = ln 2 (sum log2 pivots + 64 log2 kappa)
            logp_d[bc] = ok ? (-0.5f * quad - 0.5f * logdet - (float)LS_DP * DKT_HALF_LOG_2PI) : qnan;
            dnoise_d[bc] = ok ? 0.5f * (tt - trk) : qnan;
            jitter_used[bc] = jit;
            info[bc] = fail_at;
        }
    }
    // ---- W'[b]: tile (i, j) and its mirror (a failed class poisons the episode's W', as in dkt_mll_f32) ----
    const brsrc Wr = mk_rsrc(Wd + (size_t)b * LS_DP * LS_DP, LS_DP * LS_DP * 4);
#pragma unroll
    for (int j = 0; j < LS_NT; ++j) {
#pragma unroll
        for (int i = 0; i <= j; ++i) {
            f32x4 v = wacc[j * (j + 1) / 2 + i];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = poison ? qnan : v[q];
            if (i < j) {
                bstore4(Wr, v, ((16 * j + c16) * LS_DP + 16 * i + g4) * 4, 0);                 // rows of tile column j: the mirror, 16 bytes per lane
#pragma unroll
                for (int q = 0; q < 4; ++q) bstore1(Wr, v[q], ((16 * i + g4 + q) * LS_DP + 16 * j + c16) * 4, 0);
            } else {
                // diagonal tile: the upper triangle and its mirror
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool keep = g4 + q <= c16;
                    bstore1(Wr, v[q], keep ? ((16 * i + g4 + q) * LS_DP + 16 * j + c16) * 4 : OOB, 0);
                    bstore1(Wr, v[q], (keep && g4 + q < c16) ? ((16 * j + c16) * LS_DP + 16 * i + g4 + q) * 4 : OOB, 0);
                }
            }
        }
    }
}

}  // namespace

extern "C" int dkt_lowrank_solve_f32(const float* A, const float* P, const float* sv, const float* noise, const float* cls_weight, float jitter0,
                                     int max_tries, float* T, float* Wd, float* logp_d, float* dnoise_d, float* jitter_used, int32_t* info,
                                     int B, int C, void* stream) {
    if (!A || !P || !sv || !noise || !T || !Wd || !logp_d || !dnoise_d || !jitter_used || !info) return DKT_ERR_BAD_ARG;
    if (B <= 0 || C <= 0 || max_tries < 0 || max_tries > 8) return DKT_ERR_BAD_ARG;
    hipLaunchKernelGGL(lowrank_solve_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, A, P, sv, noise, cls_weight, jitter0, max_tries, T, Wd, logp_d,
                       dnoise_d, jitter_used, info, C);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
