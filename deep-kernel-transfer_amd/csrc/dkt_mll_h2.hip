// dkt_mll_h2.hip -- exact-GP marginal likelihood for N + 1 <= 128, ONE WAVE PER CLASS MATRIX (the structure of dkt_mll_mfma.hip),
// with the factorisation and the K^-1 product on the f16 MATRIX PIPE: v_mfma_f32_16x16x16_f16 on scaled 2-way f16 splits.
//
// Replaces `-self.mll(output, targets)`, its autograd backward and the eval-mode mean cache
// (reference methods/DKT.py:161-163, 177, 187, 252-254, 265, 330; methods/DKT_regression.py:53-56, 92), i.e. GPyTorch's
// psd_safe_cholesky / inv_quad_logdet / cholesky_solve, for the C one-vs-rest models K_c = sv_c E + noise_c I of an episode.
//
// Why: on gfx950 v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate and shares its issue pipe with the VALU work of the
// diagonal-tile sweeps; the f16 / bf16 MFMAs run at 16 x that rate on a pipe of their own.  A tile in the
// accumulator layout (lane (g, c), register q  <->  element [4g + q][c]) is, with its four registers converted to f16 and packed
// into two VGPRs, a legal A operand of v_mfma_f32_16x16x16_f16 (A[c][4g + e] = X[4g + e][c]: A = X^T) AND a legal B operand
// (B[4g + e][c] = Y[4g + e][c]) -- so  D += X^T Y  is ONE instruction per pair of f16 planes, no LDS, no shuffles.  A stored
// operand tile is the pair (h, m) = (f16(x s), f16(x s - h)), s a power of two: 4 VGPRs, exactly the footprint of the fp32
// tile it replaces, 22 significand bits, and  x^T y = (hh + hm + mh) / (s_x s_y)  to 2^-22 relative (the m m term is dropped):
// three f16 MFMAs (48 cycles of the matrix pipe) instead of four fp32 MFMAs (128 cycles of the shared pipe) per tile product.
//
// Scales (all exact powers of two; f16 is a FLOATING format, so an element keeps its 22 bits as long as it lies within 2^-18
// of its bound):
//   * phase 1 (factorisation K' = R^T R, right-looking):  K is scaled by kappa = 4^m >= max K_ii, hence every Schur
//     complement entry and every entry of R is bounded by 1 -> s = 2^15, rigorous and tight.  The accumulators hold 2^30 S.
//     The augmented column r is scaled by rho = 2^-e <= 1 / (|r_s| mu2), mu2 >= sqrt(kappa / (noise + jitter)) >= |R^-1|,
//     which bounds w = R^-T r (and every intermediate of that column) by 1.  The inverse of a diagonal tile, M_kk, enters
//     the panel product with a scale read off its own largest element.
//   * phase 2 (M = R^-T) stays on the fp32 matrix instruction: M has no a-priori bound that is tight enough for a static
//     scale (measured with the lane-level model, tools/mll_mfma_model.py run_h2: a loose bound costs 3 digits of alpha).
//     R is re-joined (h + m, one v_fma_mix_f32 per element) as each block column is consumed.
//   * phase 3 (P = M^T M = K_s^-1): ONE scale for all tiles of M, read off max |M| after phase 2 -- exact, so no bound is
//     involved.  The augmented row of M (= -rho alpha_s^T) is zeroed first; the rank-one term -alpha alpha^T of
//     d logp / d K is added in fp32 when the class contributions are summed (it used to ride on the product through a sign
//     flip, which the rho scaling of that row no longer allows).
//
// Diagonal tiles, E staging, the sum over the classes in LDS and the stores are those of dkt_mll_mfma.hip, which stays in the
// library as the exact-fp32 twin (flag DKT_MLL_FORCE_F32MFMA) and serves DKT_MLL_WANT_CHOL.
#include "dkt_h2_tiles.h"
#include <cstdlib>

namespace {

using namespace dkt_mfma;

// Phase 2 of the wave-per-episode kernel on the f16 pipe too (1, default) or on v_mfma_f32_16x16x4_f32 with M re-split for phase 3 (0)
#ifndef DKT_H2E_P2H
#define DKT_H2E_P2H 1
#endif
#ifndef DKT_H2_SWEEP_PRIO
#define DKT_H2_SWEEP_PRIO 1
#endif
#define DKT_SWEEP_PRIO(p) do { if (DKT_H2_SWEEP_PRIO) __builtin_amdgcn_s_setprio(p); } while (0)

constexpr int ntt(int nt) { return nt * (nt + 1) / 2; }
__host__ __device__ constexpr int tidx(int i, int j) { return j * (j + 1) / 2 + i; }      // i <= j

constexpr int H2_MAX_WPG = 5;

#ifdef DKT_MFMA_CLOCKS      // measurement build (tools/mll_phase_clocks.py): s_memtime stamps per wave into the workspace pointer
#define DKT_CLK(i) do { __builtin_amdgcn_sched_barrier(0); clk[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DKT_CLK(i) do { } while (0)
#endif

template <int NT>
struct Tiles {
    f32x4 t[NT][NT];         // [i][j], i < j: off-diagonal slots;  [j][j]: diagonal slot
};

template <bool ER>
struct FormCtxT {
    static constexpr bool er = ER;   // true: the raw E tile sits in the tile's own register slot (wave-per-episode kernel); false: staged in LDS
    const f32x4* es;         // LDS: the episode's E tiles (raw, accumulator layout), shared by the waves of the episode
    const f32x4* ys;         // LDS: this wave's targets y_c, 16-byte groups
    int pN, c16, g4, lane;
    float nsv, dg, mc, rsc;  // 2^30 x: -sv / kappa, -(noise + jitter) / kappa;  mean;  2^30 rho / sqrt(kappa)
};

// Tile (I, J), I <= J, of 2^30 S = -2^30 K' in the accumulator layout, from the staged E tile.  The last block column carries the
// augmented column -rho r_s (lanes c == pN), the last diagonal tile also its mirror row, a zero at the augmented pivot and -1 on the
// padding diagonal.
template <int NT, int I, int J, class F>
__device__ __forceinline__ f32x4 form_tile(const Tiles<NT>& T, const F& f) {
    const int pN = f.pN, c16 = f.c16, g4 = f.g4;
    f32x4 e;
    if constexpr (F::er) e = T.t[I][J];
    else e = f.es[tidx(I, J) * 64 + f.lane];
    f32x4 s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = f.nsv * e[q];
        if (I == J) v = (g4 + q == c16) ? v + f.dg : v;
        s[q] = v;
    }
    if constexpr (J == NT - 1) {
        const f32x4 yv = f.ys[4 * I + (g4 >> 2)];                             // y[16 I + 4g + q]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool rok = (I < NT - 1) || (g4 + q < pN);
            s[q] = (c16 == pN) ? (rok ? (f.mc - yv[q]) * f.rsc : 0.f) : s[q];
        }
        if constexpr (I == NT - 1) {
            const float yc = reinterpret_cast<const float*>(f.ys)[16 * I + c16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = s[q];
                v = (g4 + q == pN) ? ((c16 < pN) ? (f.mc - yc) * f.rsc : 0.f) : v;   // mirror row of the augmented column; pivot N = 0
                v = (g4 + q > pN) ? ((g4 + q == c16) ? -TWO30 : 0.f) : v;       // padding: identity
                s[q] = v;
            }
        }
    }
    return s;
}

template <int NT, int J, class F>
__device__ __forceinline__ void form_row0(Tiles<NT>& T, const F& f) {      // tiles (0, J), J = 0 .. NT-1
    if constexpr (J < NT) {
        T.t[0][J] = form_tile<NT, 0, J>(T, f);
        form_row0<NT, J + 1>(T, f);
    }
}

// tile row 1 of block step 0's trailing update, the freshly formed tiles as C operands: 2^30 S_1j = form(1, j) + (2^15 R_01)^T (2^15 R_0j)
template <int NT, int J, class F>
__device__ __forceinline__ void form_trailing_row1(Tiles<NT>& T, const F& f) {
    if constexpr (J < NT) {
        T.t[1][J] = xtyh(T.t[0][1], T.t[0][J], form_tile<NT, 1, J>(T, f));
        form_trailing_row1<NT, J + 1>(T, f);
    }
}

// Trailing updates of block step K that are NOT needed by the next sweep (tile rows i >= K + 2), u = 0 .. n_pending - 1 in (i, j) order.
template <int NT, int K> constexpr int n_pending() { return (K >= 0 && NT - K - 2 > 0) ? (NT - K - 2) * (NT - K - 1) / 2 : 0; }
template <int NT, int K> constexpr int pend_i(int u) { int i = K + 2; while (u >= NT - i) { u -= NT - i; ++i; } return i; }
template <int NT, int K> constexpr int pend_j(int u) { int i = K + 2; while (u >= NT - i) { u -= NT - i; ++i; } return i + u; }

// The pending updates as a stream of single MFMAs, S = 0 .. n_pend_mfma - 1: two tile updates advance alternately, three plane
// products each.
template <int NT, int K> constexpr int n_pend_mfma() { return 6 * ((n_pending<NT, K>() + 1) / 2); }

template <int NT, int K, int S0, int S1, class F>
__device__ __forceinline__ void pend_mfma(Tiles<NT>& T, const F& f) {
    if constexpr (S0 < S1) {
        constexpr int u = 2 * (S0 / 6) + (S0 & 1), w = (S0 % 6) >> 1;
        if constexpr (u < n_pending<NT, K>()) {
            constexpr int i = pend_i<NT, K>(u), j = pend_j<NT, K>(u);
            if constexpr (K == 0 && w == 0) T.t[i][j] = form_tile<NT, i, j>(T, f);              // block step 0 consumes E as it goes
            T.t[i][j] = xtyh1<w>(T.t[K][i], T.t[K][j], T.t[i][j]);
        }
        pend_mfma<NT, K, S0 + 1, S1>(T, f);
    }
}

constexpr int slots_of_pivot(int p) { return 1 + (15 - p + 4) / 5; }
constexpr int slots_before(int p) { int n = 0; for (int i = 0; i < p; ++i) n += slots_of_pivot(i); return n; }
constexpr int SWEEP_SLOTS = slots_before(16);

template <int NT, int K, int SLOT, class F>
__device__ __forceinline__ void run_slot(Tiles<NT>& T, const F& f) {
    constexpr int NM = n_pend_mfma<NT, K>();
    if constexpr (NM > 0) {
        pend_mfma<NT, K, SLOT * NM / SWEEP_SLOTS, (SLOT + 1) * NM / SWEEP_SLOTS>(T, f);
        __builtin_amdgcn_sched_barrier(0);                    // pin the MFMA between the VALU pieces
    }
}

template <int NT, int K, int P, int I0, int SLOT, class F>
__device__ __forceinline__ void sweep_rows_slots(Tiles<NT>& T, const F& f, float (&x)[16], const float t) {
    if constexpr (I0 < 16) {
        constexpr int CNT = (16 - I0) < 5 ? (16 - I0) : 5;
        sweep_rows_piece<P, I0, CNT>(x, t);
        run_slot<NT, K, SLOT>(T, f);
        sweep_rows_slots<NT, K, P, I0 + CNT, SLOT + 1>(T, f, x, t);
    }
}

template <int NT, int K, int P, bool LAST, class F>
__device__ __forceinline__ void sweep_interleaved(Tiles<NT>& T, const F& f, float (&x)[16], float& dv, const Lane& ln, const int pn) {
    if constexpr (P < 16) {
        const float t = sweep_pivot_head<P, LAST>(x, dv, ln, pn);
        run_slot<NT, K, slots_before(P)>(T, f);
        sweep_rows_slots<NT, K, P, P + 1, slots_before(P) + 1>(T, f, x, t);
        sweep_interleaved<NT, K, P + 1, LAST>(T, f, x, dv, ln, pn);
    }
}

// accumulator layout (2^30 S) -> replicated column layout (S)
__device__ __forceinline__ void sweep_begin30(const f32x4 S, float (&x)[16], float& dv) {
#pragma unroll
    for (int q = 0; q < 4; ++q) spread_rows(S[q] * TWOM30, x[q], x[4 + q], x[8 + q], x[12 + q]);
    dv = 1.0f;
}

struct P1Ctx {
#ifdef DKT_MFMA_CLOCKS
    unsigned long long* clk;
#endif
    f32x4* myst;
    h4 negIh;
    int lane, c16, pN;
    int fail_at;
    float lsum, quad;
};

// Block step K of the factorisation; x / dv hold the swept diagonal tile K on entry.
template <int NT, int K, class F>
__device__ __forceinline__ void phase1_step(Tiles<NT>& T, const F& f, P1Ctx& c, float (&x)[16], float& dv, const Lane& ln) {
    if constexpr (K < NT) {
#ifdef DKT_MFMA_CLOCKS
        __builtin_amdgcn_sched_barrier(0); c.clk[13 + 2 * K] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);     // sweep K done
#endif
        const f32x4 M = sweep_end(x, ln);
        const bool valid = (K < NT - 1) || (c.c16 < c.pN);
        const unsigned long long badm = __ballot(valid && !(dv > 0.f)) & 0xffffull;
        const int first = (int)__builtin_ctzll(badm | 0x10000ull);
        c.fail_at = (c.fail_at == 0 && badm != 0) ? 16 * K + first + 1 : c.fail_at;
        c.lsum += (valid && ln.g0) ? __builtin_amdgcn_logf(dv) : 0.f;              // log2
        if constexpr (K == NT - 1) c.quad = -__int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv), c.pN));
        if constexpr (F::er) T.t[K][K] = M;                     // wave-per-episode kernel: M_KK takes the (dead) diagonal slot
        else c.myst[K * 64 + c.lane] = M;
        if constexpr (K + 1 < NT) {
            // the panel: R_Kj = -M_KK S_Kj = (-V_KK)^T S_Kj with V = M^T; M_KK at a scale read off its largest element (>= 1: pivots <= 1)
            float mx = fmaxf(fmaxf(fabsf(M[0]), fabsf(M[1])), fmaxf(fabsf(M[2]), fabsf(M[3])));
            mx = wave_reduce_dpp<true>(mx);
            mx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mx)));
            float sig_inv;
            const float sig = scale_for(mx, sig_inv);
            const f32x4 nV = neg_transpose_h2(split_h2(M, sig), c.negIh);                       // sig (-V_KK), split
#pragma unroll
            for (int j = K + 1; j < NT; ++j)
                T.t[K][j] = split_h2(xtyh0(nV, split_h2(T.t[K][j], TWOM15)), sig_inv);          // (sig 2^15 R_Kj) / sig -> 2^15 R_Kj, split
            // tile row K + 1 first: the next sweep and the next panel need it
            if constexpr (K == 0) form_trailing_row1<NT, 1>(T, f);
            else {
#pragma unroll
                for (int j = K + 1; j < NT; ++j) T.t[K + 1][j] = xtyh(T.t[K][K + 1], T.t[K][j], T.t[K + 1][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef DKT_MFMA_CLOCKS
            c.clk[14 + 2 * K] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);      // panel + tile row K + 1 issued
#endif
            DKT_SWEEP_PRIO(DKT_H2_SWEEP_PRIO);
            sweep_begin30(T.t[K + 1][K + 1], x, dv);
            sweep_interleaved<NT, K, 0, K + 1 == NT - 1>(T, f, x, dv, ln, c.pN);
            DKT_SWEEP_PRIO(0);
            phase1_step<NT, K + 1>(T, f, c, x, dv, ln);
        }
    }
}

// E[b] tile (I, J) (raw) for the stage: E is symmetric, so element [4g+q][c] = E[16J + c][16I + 4g + q] -- one 16-byte load per
// lane; rows / columns beyond N read as 0.
template <int NT, int I, int J>
__device__ __forceinline__ f32x4 load_e_tile(const brsrc Er, const int N, const int pN, const int c16, const int g4) {
    const int row = 16 * J + c16;
    const bool row_ok = (J < NT - 1) || (c16 < pN);
    f32x4 e;
    if constexpr (I < NT - 1) {
        e = bload4(Er, row_ok ? (row * N + g4) * 4 : OOB, 16 * I * 4);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            e[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Er, (row_ok && g4 + q < pN) ? (row * N + g4 + q) * 4 : OOB, 16 * I * 4, 0));
    }
    return e;
}

template <int NT, int I, int J>
__device__ __forceinline__ void stage_e(f32x4* es, const brsrc Er, const int N, const int pN, const int c16, const int g4, const int lane,
                                        const int w, const int wpg) {
    if constexpr (J < NT) {
        if ((tidx(I, J) % wpg) == w) es[tidx(I, J) * 64 + lane] = load_e_tile<NT, I, J>(Er, N, pN, c16, g4);
        if constexpr (I < J) stage_e<NT, I + 1, J>(es, Er, N, pN, c16, g4, lane, w, wpg);
        else stage_e<NT, 0, J + 1>(es, Er, N, pN, c16, g4, lane, w, wpg);
    }
}

// E[b] -> the tile registers, diagonal tiles first (wave-per-episode kernel)
template <int NT, int J>
__device__ __forceinline__ void load_e_diag(Tiles<NT>& T, const brsrc Er, const int N, const int pN, const int c16, const int g4) {
    if constexpr (J < NT) {
        T.t[J][J] = load_e_tile<NT, J, J>(Er, N, pN, c16, g4);
        load_e_diag<NT, J + 1>(T, Er, N, pN, c16, g4);
    }
}
template <int NT, int I, int J>
__device__ __forceinline__ void load_e_off(Tiles<NT>& T, const brsrc Er, const int N, const int pN, const int c16, const int g4) {
    if constexpr (I < NT - 1) {
        if constexpr (J < NT) {
            T.t[I][J] = load_e_tile<NT, I, J>(Er, N, pN, c16, g4);
            load_e_off<NT, I, J + 1>(T, Er, N, pN, c16, g4);
        } else {
            load_e_off<NT, I + 1, I + 2>(T, Er, N, pN, c16, g4);
        }
    }
}
template <int NT>
__device__ __forceinline__ void load_all_e(Tiles<NT>& T, const brsrc Er, const int N, const int pN, const int c16, const int g4) {
    load_e_diag<NT, 0>(T, Er, N, pN, c16, g4);
    load_e_off<NT, 0, 1>(T, Er, N, pN, c16, g4);                 // row-major over the upper triangle: the order the factorisation consumes them
}

// Episodes per workgroup: as dkt_mll_mfma.hip (two episodes = 10 waves load the SIMDs (3, 3, 2, 2) at 168 VGPRs; NT = 8: one).
template <int NT> constexpr int h2_epw() { return NT <= 7 ? 2 : 1; }

#define DKT_OPAQUE_V(x) asm volatile("" : "+v"(x))
#define DKT_OPAQUE_S(x) asm volatile("" : "+s"(x))

template <int NT, bool GRAD, bool WPG5>
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64 * H2_MAX_WPG * h2_epw<NT>()), amdgpu_waves_per_eu(NT <= 7 ? 3 : 2, NT <= 7 ? 3 : 2)))
void mll_h2_kernel(MllArgs a, const int wpg) {
    constexpr int NTT = ntt(NT);
    constexpr int EPW = h2_epw<NT>();
    __shared__ f32x4 stage_all[EPW][(GRAD && WPG5 ? 5 * ((NTT + 4) / 5) : NTT) * 64];
    __shared__ f32x4 mst[H2_MAX_WPG * EPW][NT * 64];             // per wave: the diagonal tiles M_kk
    __shared__ f32x4 yst[H2_MAX_WPG * EPW][NT * 4];              // per wave: the targets of its class; later alpha

    const int tid = threadIdx.x;
    const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave in the workgroup
    const int epl = wall / wpg;                                  // episode within the workgroup
    const int w = wall - epl * wpg;                              // wave within the episode
    const int C = a.C;
    // DKT_MLL_E_PER_CLASS (launched with wpg = 1: NT = 8, the sizes the wave-per-episode kernel does not serve): a "unit" is one (episode, class)
    // matrix with its own E[b, c] and W[b, c] -- one round, class c_fix
    const bool epc = (a.flags & DKT_MLL_E_PER_CLASS) != 0;
    const int nunits = epc ? a.B * C : a.B;
    const int unit = min((int)blockIdx.x * EPW + epl, nunits - 1);
    const int b = epc ? unit / C : unit, c_fix = epc ? unit % C : 0;
    const bool ep_ok = blockIdx.x * EPW + epl < nunits;          // an odd tail: the surplus waves only keep the barriers company
    f32x4* const stage = stage_all[epl];
    f32x4* const myst = mst[wall];
    f32x4* const myys = yst[wall];
    const float qnan = __int_as_float(0x7fc00000);
    const int nrounds = epc ? 1 : (C + wpg - 1) / wpg;

#ifdef DKT_MFMA_CLOCKS
    unsigned long long clk[32] = {};
#endif
    DKT_CLK(0);
    for (int round = 0; round < nrounds; ++round) {
        int tq = tid;
        DKT_OPAQUE_V(tq);
        int c16 = tq & 15, g4 = (tq >> 2) & 12, lane = tq & 63, N = a.N;
        DKT_OPAQUE_S(N);
        Lane ln;
        ln.lane = lane; ln.g = g4 >> 2; ln.c = c16;
        ln.g0 = ln.g == 0; ln.g1 = ln.g == 1; ln.g2 = ln.g == 2;
        const int pN = N - 16 * (NT - 1);                        // local index of the augmented row / column in the last tile
        const int c = epc ? c_fix : round * wpg + w;
        const bool active = ep_ok && c < C;
        // ---- stage E[b] (the waves of the episode share the loads) and this wave's targets ----
        if (ep_ok) {
            const brsrc Er = mk_rsrc(a.E + (size_t)unit * N * N, (unsigned)(N * N * 4));
            stage_e<NT, 0, 0>(stage, Er, N, pN, c16, g4, lane, w, wpg);
        }
        float r2 = 0.f;                                          // |y - m|^2, this lane's share
        if (active) {
            const brsrc yr = mk_rsrc(a.Y + (size_t)b * a.y_bstride + (size_t)c * N, (unsigned)(N * 4));
            float* ysf = reinterpret_cast<float*>(myys);
            const float mcv = a.mean[c];
            if (lane < 16 * NT) {
                const float yv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, lane * 4, 0, 0));     // beyond N: 0
                ysf[lane] = yv;
                r2 = (lane < N) ? (yv - mcv) * (yv - mcv) : 0.f;
            }
            if (NT > 4 && lane + 64 < 16 * NT) {
                const float yv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, (lane + 64) * 4, 0, 0));
                ysf[lane + 64] = yv;
                r2 += (lane + 64 < N) ? (yv - mcv) * (yv - mcv) : 0.f;
            }
        }
        __syncthreads();
        DKT_CLK(1);
        Tiles<NT> T;
        float coefK = 0.f, coefA = 0.f;
        if (active) {
            h4 negIh;
#pragma unroll
            for (int q = 0; q < 4; ++q) negIh[q] = (g4 + q == c16) ? (_Float16)-1.0f : (_Float16)0.0f;
            const float svc = a.sv[c], mc = a.mean[c], nzc = a.noise[c];
            const size_t bc = (size_t)b * C + c;
            FormCtxT<false> f;
            f.es = stage; f.ys = myys;
            f.pN = pN; f.c16 = c16; f.g4 = g4; f.lane = lane; f.mc = mc;
            // kappa = 4^m >= max_i K_ii (exact in fp32): every pivot of K / kappa is <= 1, every entry of K / kappa, of its Schur
            // complements and of R is bounded by 1.
            float emax = 0.f, etr = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 e = stage[tidx(j, j) * 64 + lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    emax = fmaxf(emax, (g4 + q == c16) ? e[q] : 0.f);
                    etr += (g4 + q == c16) ? e[q] : 0.f;
                }
            }
            emax = wave_reduce_dpp<true>(emax);
            // kappa-aware dispatch (dkt_mll.hip, mll_kappa_fixup): the a-priori condition bound 1 + sv trace(E) / noise from the diagonal that is in registers anyway
            const bool kappa_high = a.kappa_max > 0.f && !(1.0f + svc * wave_reduce_dpp<false>(etr) / nzc <= a.kappa_max);
            r2 = wave_reduce_dpp<false>(r2);
            int fail_at = 0;
            float jit = 0.f, lsum = 0.f, quad = 0.f, aug_unscale = 1.f;
            int msc = 0;
            for (int attempt = 0; attempt <= a.max_tries; ++attempt) {
                jit = 0.f;
                if (attempt > 0) {
                    jit = a.jitter0;
                    for (int i = 1; i < attempt; ++i) jit *= 10.f;
                }
                int ex;
                (void)frexpf(fmaf(svc, emax, nzc + jit), &ex);                 // max K_ii = f 2^ex, 0.5 <= f < 1
                msc = max(0, (ex + 1) >> 1);
                const float ikap = ldexpf(1.0f, -2 * msc);                      // 1 / kappa
                // mu2 = 2^emu >= sqrt(kappa / (noise + jitter)) >= |R^-1|;  rho = 2^-er <= 1 / (|r_s| mu2): the augmented column
                // w = R^-T r_s rho (and every intermediate of it) is bounded by 1
                const float lam = fmaxf((nzc + jit) * ikap, 9.094947017729282e-13f);         // >= 2^-40
                int exm;
                (void)frexpf(1.0f / lam, &exm);
                const int emu = (exm + 1) >> 1;
                int exr;
                (void)frexpf(r2 * ikap * ldexpf(1.0f, 2 * emu), &exr);
                const int er = (r2 > 0.f) ? ((exr + 1) >> 1) : 0;
                f.nsv = -svc * ikap * TWO30;
                f.dg = -(nzc + jit) * ikap * TWO30;
                f.rsc = ldexpf(TWO30, -msc - er);
                aug_unscale = ldexpf(1.0f, er);                                 // 1 / rho
                form_row0<NT, 0>(T, f);
                // ---- phase 1: factorisation.  The sweep of tile k + 1 runs ahead, interleaved with step k's remaining updates ----
                P1Ctx pc;
                pc.myst = myst; pc.negIh = negIh;
                pc.lane = lane; pc.c16 = c16; pc.pN = pN;
                pc.fail_at = 0; pc.lsum = 0.f; pc.quad = 0.f;
#ifdef DKT_MFMA_CLOCKS
                pc.clk = clk;
                clk[9] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
                clk[12] = __builtin_amdgcn_s_memtime();
#endif
                {
                    float dv, x[16];
                    DKT_SWEEP_PRIO(DKT_H2_SWEEP_PRIO);
                    sweep_begin30(T.t[0][0], x, dv);
                    sweep_interleaved<NT, -1, 0, NT == 1>(T, f, x, dv, ln, pN);
                    DKT_SWEEP_PRIO(0);
                    phase1_step<NT, 0>(T, f, pc, x, dv, ln);
                }
                fail_at = pc.fail_at; lsum = pc.lsum; quad = pc.quad * aug_unscale * aug_unscale;
                if (fail_at == 0) break;
            }
            DKT_CLK(2);
            // ---- phase 2 (fp32 matrix instruction): M = R^-T; M_ji (j > i) overwrites slot (i, j); two rows i at a time.
            // Column j of R is re-joined to fp32 (2^15 R) as row j starts; the 2^-15 rides on the identity of the -V_jj product. ----
            tq = tid;
            DKT_OPAQUE_V(tq);
            c16 = tq & 15; g4 = (tq >> 2) & 12; lane = tq & 63;
            f32x4 negI2;
#pragma unroll
            for (int q = 0; q < 4; ++q) negI2[q] = (g4 + q == c16) ? -TWOM15 : 0.0f;
#pragma unroll
            for (int j = 1; j < NT; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 nV = xty0(myst[j * 64 + lane], negI2);
#pragma unroll
                for (int k = 0; k < j; ++k) T.t[k][j] = join_h2(T.t[k][j]);
#pragma unroll
                for (int i = 0; i < j; i += 2) {
                    if (i + 1 < j) {
                        f32x4 QA = xty0(T.t[i][j], myst[i * 64 + lane]);                 // k = i
                        f32x4 QB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = i + 1; k < j; ++k)
                            xty2(T.t[k][j], T.t[i][k], QA, T.t[k][j], (k == i + 1) ? myst[k * 64 + lane] : T.t[i + 1][k], QB);
                        f32x4 RA = {0.f, 0.f, 0.f, 0.f}, RB = {0.f, 0.f, 0.f, 0.f};
                        xty2(nV, QA, RA, nV, QB, RB);
                        T.t[i][j] = RA;
                        T.t[i + 1][j] = RB;
                    } else {
                        f32x4 Q = xty0(T.t[i][j], myst[i * 64 + lane]);
#pragma unroll
                        for (int k = i + 1; k < j; ++k) Q = xty(T.t[k][j], T.t[i][k], Q);
                        T.t[i][j] = xty0(nV, Q);
                    }
                }
            }
            DKT_CLK(3);
            // ---- alpha = -(row N of M) / rho / sqrt(kappa);  kept in LDS (the targets are dead) for the rank-one term ----
            const bool arow = (ln.g == (pN >> 2));
            const int qn = pN & 3;
            float asum = 0.f, aa = 0.f;
            {
                const brsrc ar = mk_rsrc(a.alpha + bc * N, (unsigned)(N * 4));
                const float asc = ldexpf(aug_unscale, -msc);
                float* ysf = reinterpret_cast<float*>(myys);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const f32x4 m = (i < NT - 1) ? T.t[i][NT - 1] : myst[(NT - 1) * 64 + lane];
                    const float v = -(qn == 0 ? m[0] : qn == 1 ? m[1] : qn == 2 ? m[2] : m[3]) * asc;
                    const bool ok = arow && ((i < NT - 1) || (c16 < pN));
                    bstore1(ar, (fail_at != 0) ? qnan : v, ok ? (16 * i + c16) * 4 : OOB, 0);
                    if (GRAD && arow) ysf[16 * i + c16] = ok ? v : 0.f;
                    asum += ok ? v : 0.f;
                    aa += ok ? v * v : 0.f;
                }
            }
            float trk = 0.f;
            DKT_CLK(4);
            if constexpr (GRAD) {
                // ---- phase 3: P_ij = sum_{k >= j} M_ki^T M_kj = K_s^-1, in place, on the f16 pipe.  One scale for all of M, read off
                // its largest element; the augmented row (-rho alpha_s^T, M_NN = 1) is zeroed first. ----
                float mx = 0.f;
#pragma unroll
                for (int i = 0; i < NT - 1; ++i) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) T.t[i][NT - 1][q] = (g4 + q == pN) ? 0.f : T.t[i][NT - 1][q];
                }
#pragma unroll
                for (int j = 1; j < NT; ++j) {
#pragma unroll
                    for (int i = 0; i < j; ++i) {
                        const f32x4 v = T.t[i][j];
                        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const f32x4 v = myst[j * 64 + lane];
#pragma unroll
                    for (int q = 0; q < 4; ++q) mx = fmaxf(mx, (j == NT - 1 && g4 + q == pN) ? 0.f : fabsf(v[q]));
                }
                mx = wave_reduce_dpp<true>(mx);
                mx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mx)));
                float s3_inv;
                const float s3 = scale_for(mx, s3_inv);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    f32x4 v = myst[j * 64 + lane];
                    if (j == NT - 1) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = (g4 + q == pN) ? 0.f : v[q];
                    }
                    myst[j * 64 + lane] = split_h2(v, s3);
                }
#pragma unroll
                for (int j = 1; j < NT; ++j) {
#pragma unroll
                    for (int i = 0; i < j; ++i) T.t[i][j] = split_h2(T.t[i][j], s3);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    __builtin_amdgcn_sched_barrier(0);
                    // i = 0 .. j-1 in pairs, then the diagonal (whose LDS slot the others still read) last
#pragma unroll
                    for (int i = 0; i <= j; i += 2) {
                        const bool pair = i + 1 <= j;
                        f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = j; k < NT; ++k) {
                            const f32x4 Bm = (k == j) ? myst[k * 64 + lane] : T.t[j][k];
                            const f32x4 A0 = (k == i) ? myst[k * 64 + lane] : T.t[i][k];
                            if (pair) {
                                const f32x4 A1 = (k == i + 1) ? myst[k * 64 + lane] : T.t[i + 1][k];
                                xtyh2(A0, Bm, accA, A1, Bm, accB);
                            } else {
                                accA = xtyh(A0, Bm, accA);
                            }
                        }
                        auto put = [&](const f32x4 acc, const int ii) {
                            if (ii < j) {
                                T.t[ii][j] = acc;
                            } else {
                                // diagonal tile: M_jj is dead now, P_jj takes its place in LDS; trace over the real rows
                                myst[j * 64 + lane] = acc;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const bool ok = (g4 + q == c16) && ((j < NT - 1) || (c16 < pN));
                                    trk += ok ? acc[q] : 0.f;
                                }
                            }
                        };
                        put(accA, i);
                        if (pair) put(accB, i + 1);
                    }
                }
                trk *= s3_inv * s3_inv * ldexpf(1.0f, -2 * msc);                    // tr K^-1
                const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
                // W_c = -0.5 cw sv (K^-1 - alpha alpha^T) = coefK P_acc + coefA alpha alpha^T   (a failed class poisons W[b])
                coefK = (fail_at == 0) ? -0.5f * cw * a.sv[c] * s3_inv * s3_inv * ldexpf(1.0f, -2 * msc) : qnan;
                coefA = 0.5f * cw * a.sv[c];
            }
            DKT_CLK(5);
            lsum = wave_reduce_dpp<false>(lsum) + (float)(2 * msc * N);            // log2 det K = log2 det K_s + N log2 kappa
            asum = wave_reduce_dpp<false>(asum);
            aa = wave_reduce_dpp<false>(aa);
            trk = wave_reduce_dpp<false>(trk);
            if (lane == 0) {
                const bool ok = fail_at == 0;
                a.logp[bc] = ok ? (-0.5f * quad - 0.34657359027997264f * lsum - (float)N * DKT_HALF_LOG_2PI) : qnan;
                a.jitter_used[bc] = jit;
                a.info[bc] = (ok && kappa_high) ? -1 : fail_at;
                if constexpr (GRAD) {
                    const float nz_eff = a.noise[c] + jit;
                    const float trpp = trk - aa;                                    // tr (K^-1 - alpha alpha^T)
                    a.dmean[bc] = ok ? asum : qnan;
                    a.dnoise[bc] = ok ? -0.5f * trpp : qnan;                                    // 0.5 (alpha.alpha - tr K^-1)
                    a.dsv[bc] = ok ? 0.5f * ((quad - (float)N) + nz_eff * trpp) / a.sv[c] : qnan;
                }
            }
        }
        DKT_CLK(6);
        if constexpr (GRAD) {
            // ---- W[b] = sum over the classes of coefK_c P_c + coefA_c alpha_c alpha_c^T.  The staged E is dead once every wave is
            // past its factorisation (the barrier); its LDS becomes the exchange buffer. ----
            __syncthreads();
            tq = tid;
            DKT_OPAQUE_V(tq);
            c16 = tq & 15; g4 = (tq >> 2) & 12; lane = tq & 63;
            DKT_OPAQUE_S(N);
            const int pNs = N - 16 * (NT - 1);
            const brsrc Wr = mk_rsrc(a.W + (size_t)unit * N * N, (unsigned)(N * N * 4));
            const float* alf = reinterpret_cast<const float*>(myys);
            // this wave's scaled tile n = (i, j): coefK P_ij + (coefA alpha_row) alpha_col
            auto scaled_tile = [&](const f32x4 p, const int i, const int j) {
                const f32x4 ar = myys[4 * i + (g4 >> 2)];                      // alpha[16 i + 4g + q]
                const float ac = alf[16 * j + c16] * coefA;
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaf(ar[q], ac, p[q] * coefK);
                return v;
            };
            // tile n -> W[b]: element [4g+q][c] of tile (i, j) and its mirror; diagonal tiles: the upper triangle and ITS mirror (the
            // hm + mh plane products are symmetric only up to the order of two roundings); later rounds add to what this wave stored
            auto store_tile = [&](f32x4 v, const int i, const int j) {
                const bool col_ok = (j < NT - 1) || (c16 < pNs);
                const int vo_m = col_ok ? ((16 * j + c16) * N + 16 * i + g4) * 4 : OOB;       // i < j <= NT-1: real columns
                int vo_d[4], vo_t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool row_ok = (i < NT - 1) || (g4 + q < pNs);
                    const bool keep = row_ok && col_ok && (i < j || g4 + q <= c16);
                    vo_d[q] = keep ? ((16 * i + g4 + q) * N + 16 * j + c16) * 4 : OOB;
                    vo_t[q] = (keep && g4 + q < c16) ? ((16 * j + c16) * N + 16 * i + g4 + q) * 4 : OOB;
                }
                if (round > 0) {
                    if (i < j) {
                        v += bload4(Wr, vo_m, 0);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Wr, vo_d[q], 0, 0));
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) bstore1(Wr, v[q], vo_d[q], 0);
                if (i < j) bstore4(Wr, v, vo_m, 0);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) bstore1(Wr, v[q], vo_t[q], 0);
                }
            };
            if constexpr (WPG5) {
                constexpr int KMAX = (NTT + 4) / 5;
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    if (ep_ok) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
#pragma unroll
                            for (int i = 0; i <= j; ++i) {
                                const int n = tidx(i, j);
                                if (n % 5 != g) continue;
                                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                                if (active) v = scaled_tile((i < j) ? T.t[i][j] : myst[j * 64 + lane], i, j);
                                stage[(w * KMAX + n / 5) * 64 + lane] = v;
                            }
                        }
                    }
                    __syncthreads();
                    if (ep_ok) {
                        for (int k = w; 5 * k + g < NTT; k += 5) {
                            f32x4 v = stage[k * 64 + lane];
#pragma unroll
                            for (int ww = 1; ww < 5; ++ww) v += stage[(ww * KMAX + k) * 64 + lane];
                            const int n = 5 * k + g;
                            int j = 0;
                            while ((j + 1) * (j + 2) / 2 <= n) ++j;          // wave-uniform: tile n = (i, j), i <= j
                            store_tile(v, n - j * (j + 1) / 2, j);
                        }
                    }
                    if (g < 4 || round + 1 < nrounds) __syncthreads();
                }
            } else {
                // general class count per round: the waves take turns adding all their tiles (fixed order), then share the stores
                for (int t = 0; t < wpg; ++t) {
                    if (t == w && active) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
#pragma unroll
                            for (int i = 0; i <= j; ++i) {
                                f32x4 v = scaled_tile((i < j) ? T.t[i][j] : myst[j * 64 + lane], i, j);
                                f32x4* p = &stage[tidx(i, j) * 64 + lane];
                                if (t > 0) v += *p;
                                *p = v;
                            }
                        }
                    }
                    __syncthreads();
                }
                if (ep_ok) {
                    for (int n = w; n < NTT; n += wpg) {
                        int j = 0;
                        while ((j + 1) * (j + 2) / 2 <= n) ++j;
                        store_tile(stage[n * 64 + lane], n - j * (j + 1) / 2, j);
                    }
                }
                if (round + 1 < nrounds) __syncthreads();
            }
        } else {
            if (round + 1 < nrounds) __syncthreads();            // the next round re-stages E
        }
    }
    DKT_CLK(7);
#ifdef DKT_MFMA_CLOCKS
    DKT_CLK(8);
    if ((tid & 63) == 0 && a.ws) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.ws) + ((size_t)blockIdx.x * (H2_MAX_WPG * EPW) + wall) * 48;
        for (int i = 0; i < 32; ++i) o[i] = clk[i];
    }
#endif
}

// ======================================================================================================================
// WAVE PER EPISODE (large batches, training call): one 64-thread workgroup = one episode, the C class matrices in sequence,
// W[b] accumulated over the classes IN THE ACCUMULATORS of the phase-3 products.
//
// Why (tools/mll_phase_clocks.py on the wave-per-matrix kernel above, 8192 cfg2 episodes): of a wave's 143 k cycles only 75 k are
// the three phases; 18 k go to the E stage + its barrier and 47 k to the end of the episode (waiting for the slowest of the ten
// waves -- the (3, 3, 2, 2) placement --, five exchange rounds through LDS with their barriers, partial-width stores), and a wave
// that has a SIMD to itself needs 47 k for the same three phases.  Here nothing is exchanged and nobody waits:
//   * no barrier, no LDS staging: a class iteration loads E[b] straight into the tile registers and forms K_c in place as each tile
//     is first used (the re-reads of an episode's 44 KB are served by L2 / the memory-side cache);
//   * the inverses of the diagonal tiles stay in the (dead) diagonal register slots instead of LDS;
//   * phase 3 accumulates P_c = (q_c M_c)^T (q_c M_c) INTO the episode's W tiles: the class weight is folded into the split scale,
//     q_c = sqrt(0.5 |cw_c| sv_c / kappa_c) 2^(15 - e) (the scale of a split need not be a power of two: x q is one fp32 rounding), so
//     the sum over the classes costs no instruction at all.  The accumulators hold sigma 2^u W, sigma = -sign(cw_c) -- a class of the
//     other sign negates them first, a class whose M needs a larger bound rescales them (both rare, wave-uniform branches).  They
//     live in LDS (19 tiles of 1 KB: what eight single-wave workgroups per CU leave to each) and, beyond that, in registers, and
//     are the C operand of their column's chains once per class;
//   * the rank-one terms -alpha alpha^T are added on the VALU per class; W[b] is stored once, at the end.
// 256 VGPRs, two waves per SIMD, every SIMD carries the same load.  NT <= 7.  Small batches keep the wave-per-matrix kernel, which
// spreads one episode over five waves.
template <int NT> constexpr int h2e_wlds() { return ntt(NT) < 19 ? ntt(NT) : 19; }      // W tiles (column-major tile index) that live in LDS

template <int NT>
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(2, 2)))
void mll_h2e_kernel(MllArgs a) {
    constexpr int WLDS = h2e_wlds<NT>();
    constexpr int WREG = ntt(NT) - WLDS;
    __shared__ f32x4 wl[WLDS * 64];                              // W accumulators, tiles 0 .. WLDS-1
    __shared__ f32x4 yst[NT * 4];                                // the targets of the class; later alpha

    // DKT_MLL_E_PER_CLASS: one workgroup per (episode, class) matrix -- its own E[b, c], its own W[b, c], a class "loop" of one
    const bool epc = (a.flags & DKT_MLL_E_PER_CLASS) != 0;
    const bool want_grad = (a.flags & DKT_MLL_WANT_GRAD) != 0;          // without it (test-time conditioning on per-class matrices): phases 1 - 2 only
    const int C = a.C;
    const int b = epc ? (int)(blockIdx.x / (unsigned)C) : (int)blockIdx.x;
    const int c_first = epc ? (int)(blockIdx.x % (unsigned)C) : 0, c_end = epc ? c_first + 1 : C;
    const size_t mat = epc ? (size_t)b * C + c_first : (size_t)b;          // index of the [N, N] matrices E / W of this workgroup
    const float qnan = __int_as_float(0x7fc00000);
    int tq = threadIdx.x;
    DKT_OPAQUE_V(tq);
    int c16 = tq & 15, g4 = (tq >> 2) & 12, lane = tq & 63, N = a.N;
    DKT_OPAQUE_S(N);

    f32x4 wr[WREG > 0 ? WREG : 1];                               // W accumulators, tiles WLDS .. ntt-1
#pragma unroll
    for (int n = 0; n < WREG; ++n) wr[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < WLDS; ++n) wl[n * 64 + lane] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef DKT_MFMA_CLOCKS
    unsigned long long ph[8] = {}, tprev = __builtin_amdgcn_s_memtime();
#define DKT_PH(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[i] += tn - tprev; tprev = tn; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DKT_PH(i) do { } while (0)
#endif
    int e_acc = -1000;                                           // the accumulators' unit is 2^(2 (15 - e_acc))
    float sigma = 0.f;                                           // their sign (0: still empty)
    bool poison = false;

    // One loop over (class, attempt): a factorisation that meets a non-positive pivot restarts the class with the next jitter of
    // psd_safe_cholesky's ladder.  (A retry loop AROUND the factorisation keeps everything it might need alive across its back edge:
    // +70 VGPRs at NT = 7, measured.)
    int attempt = 0;
    for (int c = c_first; c < c_end;) {
        // lane coordinates and sizes made opaque per class (and again per phase): keeps the compiler from hoisting -- and then spilling --
        // every mask and address of the loop body into the kernel prologue
        tq = threadIdx.x;
        DKT_OPAQUE_V(tq);
        c16 = tq & 15; g4 = (tq >> 2) & 12; lane = tq & 63; N = a.N;
        DKT_OPAQUE_S(N);
        const int pN = N - 16 * (NT - 1);
        Lane ln;
        ln.lane = lane; ln.g = g4 >> 2; ln.c = c16;
        ln.g0 = ln.g == 0; ln.g1 = ln.g == 1; ln.g2 = ln.g == 2;
        const float svc = a.sv[c], mc = a.mean[c], nzc = a.noise[c];
        const size_t bc = (size_t)b * C + c;
        // ---- targets -> LDS, |y - m|^2 ----
        float r2 = 0.f;
        {
            const brsrc yr = mk_rsrc(a.Y + (size_t)b * a.y_bstride + (size_t)c * N, (unsigned)(N * 4));
            float* ysf = reinterpret_cast<float*>(yst);
            {
                const float yv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, lane * 4, 0, 0));     // beyond N: 0
                if (lane < 16 * NT) ysf[lane] = yv;
                r2 = (lane < N) ? (yv - mc) * (yv - mc) : 0.f;
            }
            if (NT > 4 && lane + 64 < 16 * NT) {
                const float yv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, (lane + 64) * 4, 0, 0));
                ysf[lane + 64] = yv;
                r2 += (lane + 64 < N) ? (yv - mc) * (yv - mc) : 0.f;
            }
        }
        r2 = wave_reduce_dpp<false>(r2);
        h4 negIh;
#pragma unroll
        for (int q = 0; q < 4; ++q) negIh[q] = (g4 + q == c16) ? (_Float16)-1.0f : (_Float16)0.0f;
        FormCtxT<true> f;
        f.es = nullptr; f.ys = yst;
        f.pN = pN; f.c16 = c16; f.g4 = g4; f.lane = lane; f.mc = mc;
        Tiles<NT> T;
        int fail_at = 0;
        bool kappa_high = false;
        float jit = 0.f, lsum = 0.f, quad = 0.f, aug_unscale = 1.f;
        int msc = 0;
        const brsrc Er = mk_rsrc(a.E + mat * N * N, (unsigned)(N * N * 4));
        {
            if (attempt > 0) {
                jit = a.jitter0;
                for (int i = 1; i < attempt; ++i) jit *= 10.f;
            }
            // ---- E[b] -> the tile registers (diagonal tiles first: kappa needs max E_ii) ----
            load_all_e<NT>(T, Er, N, pN, c16, g4);
            float emax = 0.f, etr = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    emax = fmaxf(emax, (g4 + q == c16) ? T.t[j][j][q] : 0.f);
                    etr += (g4 + q == c16) ? T.t[j][j][q] : 0.f;
                }
            }
            emax = wave_reduce_dpp<true>(emax);
            // kappa-aware dispatch (dkt_mll.hip, mll_kappa_fixup): the a-priori condition bound 1 + sv trace(E) / noise from the diagonal that is in registers anyway
            kappa_high = a.kappa_max > 0.f && !(1.0f + svc * wave_reduce_dpp<false>(etr) / nzc <= a.kappa_max);
            int ex;
            (void)frexpf(fmaf(svc, emax, nzc + jit), &ex);                 // max K_ii = f 2^ex, 0.5 <= f < 1
            msc = max(0, (ex + 1) >> 1);
            const float ikap = ldexpf(1.0f, -2 * msc);                      // 1 / kappa
            const float lam = fmaxf((nzc + jit) * ikap, 9.094947017729282e-13f);         // >= 2^-40
            int exm;
            (void)frexpf(1.0f / lam, &exm);
            const int emu = (exm + 1) >> 1;
            int exr;
            (void)frexpf(r2 * ikap * ldexpf(1.0f, 2 * emu), &exr);
            const int er = (r2 > 0.f) ? ((exr + 1) >> 1) : 0;
            f.nsv = -svc * ikap * TWO30;
            f.dg = -(nzc + jit) * ikap * TWO30;
            f.rsc = ldexpf(TWO30, -msc - er);
            aug_unscale = ldexpf(1.0f, er);                                 // 1 / rho
            form_row0<NT, 0>(T, f);
            DKT_PH(0);
            P1Ctx pc;
            pc.myst = nullptr; pc.negIh = negIh;
            pc.lane = lane; pc.c16 = c16; pc.pN = pN;
            pc.fail_at = 0; pc.lsum = 0.f; pc.quad = 0.f;
#ifdef DKT_MFMA_CLOCKS
            unsigned long long clkdummy[32];
            pc.clk = clkdummy;
#endif
            {
                float dv, x[16];
                sweep_begin30(T.t[0][0], x, dv);
                sweep_interleaved<NT, -1, 0, NT == 1>(T, f, x, dv, ln, pN);
                phase1_step<NT, 0>(T, f, pc, x, dv, ln);
            }
            fail_at = pc.fail_at; lsum = pc.lsum; quad = pc.quad * aug_unscale * aug_unscale;
        }
        DKT_PH(1);
        if (fail_at != 0 && attempt < a.max_tries) {
            ++attempt;
            continue;                                                   // same class, next jitter
        }
        tq = threadIdx.x;
        DKT_OPAQUE_V(tq);
        c16 = tq & 15; g4 = (tq >> 2) & 12; lane = tq & 63;
#if DKT_H2E_P2H
        // ---- phase 2 on the f16 pipe: M~ = g_c R^-T with g_c = sqrt(0.5 |cw_c| sv_c / kappa_c) folded in, so that phase 3 -- (M~)^T M~ =
        // |coefficient of K_c^-1 in W| K_c^-1 -- accumulates straight into the episode's W tiles.  Every finished tile of M~ is stored
        // as an f16 split at ONE scale 2^(15 - e2) for the whole matrix: e2 starts from the largest element of the diagonal tiles (all
        // known after phase 1) plus a guard binade and GROWS when a tile row comes out larger -- the stored tiles are then re-scaled by
        // an exact power of two (v_pk_mul_f16), a wave-uniform, rare branch.  M_ji = (-V_jj)^T Q_ji, Q_ji = sum_k R_kj^T M~_ki: R is used
        // as it was stored in phase 1 (2^15 R), the diagonal tiles serve twice -- without g_c as -V_jj (split on the fly), with g_c as
        // M~_ii -- and Q is split 2^-8 below M~'s scale (|Q| <= 4 max |M~_ji|: only a row that comes out 128 x above the scale overflows
        // the f16 range -- reported as a failed matrix, loudly).
        const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
        const float wmag = 0.5f * fabsf(cw) * svc;                          // |coefficient| of (K^-1 - alpha alpha^T) in W
        const float gsc = (wmag > 0.f) ? __builtin_sqrtf(wmag) * ldexpf(1.0f, -msc) : ldexpf(1.0f, -msc);       // g_c (a zero-weight class: any value)
        const bool arow = (ln.g == (pN >> 2));
        const int qn = pN & 3;
        const float asc = ldexpf(aug_unscale, -msc);
        float asum = 0.f, aa = 0.f;
        const brsrc ar = mk_rsrc(a.alpha + bc * N, (unsigned)(N * 4));
        float* ysf = reinterpret_cast<float*>(yst);
        auto emit_alpha = [&](const float v, const int i) {               // alpha[16 i + c16] = v on the lanes of row N's row group
            const bool ok = arow && ((i < NT - 1) || (c16 < pN));
            bstore1(ar, (fail_at != 0) ? qnan : v, ok ? (16 * i + c16) * 4 : OOB, 0);
            if (arow) ysf[16 * i + c16] = ok ? v : 0.f;
            asum += ok ? v : 0.f;
            aa += ok ? v * v : 0.f;
        };
        {   // the last segment of alpha = -(row N of M_dd) / rho / sqrt(kappa): from the fp32 diagonal tile, before it is split
            const f32x4 m = T.t[NT - 1][NT - 1];
            emit_alpha(-(qn == 0 ? m[0] : qn == 1 ? m[1] : qn == 2 ? m[2] : m[3]) * asc, NT - 1);
        }
        float mx = 0.f, mxt0 = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const f32x4 v = T.t[j][j];
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            if (j == 0) mxt0 = mx;
        }
        mx = wave_reduce_dpp<true>(mx);
        const float mx0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mx)));       // max |M_kk|: the -V_jj operands (no g_c) take their scale from it
        // validation aid (DKT_MLL_P2_GUARD < 0): start from the FIRST diagonal tile only, so that ordinary matrices -- whose M grows along
        // the diagonal -- exercise the grow-on-demand path row after row
        if (a.p2_guard < 0) mx = wave_reduce_dpp<true>(mxt0);
        mx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mx * gsc)));
        const bool sane = (fail_at == 0) && (mx > 0.f) && (mx < 3.0e38f) && (mx0 < 3.0e38f);
        const float unitX = sane ? ldexpf(1.0f, 15 - ((int)((__float_as_uint(mx0) >> 23) & 0xffu) - 126)) : 1.0f;      // unitX max |M_kk| < 2^15
        int e2 = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 126 + max(a.p2_guard, 0);      // g_c max |M_kk| = f 2^e, 0.5 <= f < 1
        if (sigma != 0.f) e2 = max(e2, e_acc);                              // never below the accumulators' unit: only THEY are re-scaled in phase 3
        e2 = sane ? min(max(e2, -100), 100) : 0;
        float unit2 = ldexpf(1.0f, 15 - e2);
        bool ovf = false;
        auto zero_aug_row = [&](f32x4 v) {                                 // the augmented row (-rho alpha_s^T, M_NN = 1) takes no part in K^-1
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (g4 + q == pN) ? 0.f : v[q];
            return v;
        };
        auto scale_planes = [&](const f32x4 raw, const _Float16 f) {      // exact: a power of two on both f16 planes
            Sp sp = as_sp(raw);
            sp.h = sp.h * f;
            sp.m = sp.m * f;
            return __builtin_bit_cast(f32x4, sp);
        };
#pragma unroll
        for (int j = 1; j < NT; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 nV = neg_transpose_h2(split_h2(T.t[j][j], unitX), negIh);             // -unitX V_jj (no g_c)
            T.t[j - 1][j - 1] = split_h2(T.t[j - 1][j - 1], gsc * unit2);                       // M~_(j-1)(j-1): its -V has been taken a row ago
            float mrow = 0.f;
#pragma unroll
            for (int i = 0; i < j; i += 2) {
                if (i + 1 < j) {
                    f32x4 QA = xtyh0(T.t[i][j], T.t[i][i]);
                    f32x4 QB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = i + 1; k < j; ++k) xtyh2(T.t[k][j], T.t[i][k], QA, T.t[k][j], T.t[i + 1][k], QB);      // (k == i + 1: the diagonal slot)
                    f32x4 RA = {0.f, 0.f, 0.f, 0.f}, RB = {0.f, 0.f, 0.f, 0.f};
                    xtyh2(nV, split_h2(QA, 1.1920928955078125e-07f), RA, nV, split_h2(QB, 1.1920928955078125e-07f), RB);     // 2^-23: Q at M~'s scale / 2^8
                    T.t[i][j] = RA;
                    T.t[i + 1][j] = RB;
                    mrow = fmaxf(mrow, fmaxf(fmaxf(fmaxf(fabsf(RA[0]), fabsf(RA[1])), fmaxf(fabsf(RA[2]), fabsf(RA[3]))),
                                             fmaxf(fmaxf(fabsf(RB[0]), fabsf(RB[1])), fmaxf(fabsf(RB[2]), fabsf(RB[3])))));
                } else {
                    f32x4 Q = xtyh0(T.t[i][j], T.t[i][i]);
#pragma unroll
                    for (int k = i + 1; k < j; ++k) Q = xtyh(T.t[k][j], T.t[i][k], Q);
                    const f32x4 RA = xtyh0(nV, split_h2(Q, 1.1920928955078125e-07f));
                    T.t[i][j] = RA;
                    mrow = fmaxf(mrow, fmaxf(fmaxf(fabsf(RA[0]), fabsf(RA[1])), fmaxf(fabsf(RA[2]), fabsf(RA[3]))));
                }
            }
            // the row's tiles hold acc = unitX (g unit2 M_ji) / 2^8; largest element of the row -- its diagonal tile included, which is
            // split a row later -- at M~'s scale:
            {
                const f32x4 v = T.t[j][j];
                const float md = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                mrow = fmaxf(mrow * (256.0f / unitX), md * gsc * unit2);
            }
            mrow = wave_reduce_dpp<true>(mrow);
            const float ymax = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mrow)));
            float fac = 256.0f / unitX;                                     // acc -> g unit2 M_ji
            ovf = ovf || !(ymax < 3.0e38f);
            if (sane && ymax >= 32768.0f && ymax < 3.0e38f) {
                // this row is larger than everything before it: grow the matrix' scale, re-scale what is stored (exact)
                const int dgrow = (int)((__float_as_uint(ymax) >> 23) & 0xffu) - 126 - 15;       // ymax 2^-dgrow < 2^15
                const _Float16 fh = (_Float16)ldexpf(1.0f, -dgrow);
#pragma unroll
                for (int k = 0; k < j; ++k) {
#pragma unroll
                    for (int i = 0; i <= k; ++i) T.t[i][k] = scale_planes(T.t[i][k], fh);
                }
                e2 += dgrow;
                unit2 = ldexpf(unit2, -dgrow);
                fac = ldexpf(fac, -dgrow);
            }
            if (j == NT - 1) {
                // alpha = -(row N of M) / rho / sqrt(kappa): M_ji = acc 2^8 / (g unitX unit2_old) = acc fac / (g unit2)
                const float aun = fac / (gsc * unit2) * asc;
#pragma unroll
                for (int i = 0; i < NT - 1; ++i) {
                    const f32x4 m = T.t[i][NT - 1];
                    emit_alpha(-(qn == 0 ? m[0] : qn == 1 ? m[1] : qn == 2 ? m[2] : m[3]) * aun, i);
                }
            }
#pragma unroll
            for (int i = 0; i < j; ++i) T.t[i][j] = split_h2((j == NT - 1) ? zero_aug_row(T.t[i][j]) : T.t[i][j], fac);
        }
        T.t[NT - 1][NT - 1] = split_h2(zero_aug_row(T.t[NT - 1][NT - 1]), gsc * unit2);
        if (ovf && fail_at == 0) fail_at = N + 1;                          // f16 range exceeded inside phase 2 (see above): the matrix is reported as failed
        DKT_PH(2);
        DKT_PH(3);
        tq = threadIdx.x;
        DKT_OPAQUE_V(tq);
        c16 = tq & 15; g4 = (tq >> 2) & 12; lane = tq & 63;
        // ---- phase 3 on the f16 pipe, accumulated into the episode's W tiles: M~ is already split at the scale 2^(15 - e2) ----
        const bool live = want_grad && (fail_at == 0) && (wmag > 0.f) && sane;     // a zero-weight class contributes nothing
        poison = poison || (fail_at != 0);
        float trk = 0.f;
        if (live) {
            const int e_c = e2;
            const float sg = (cw > 0.f) ? -1.0f : 1.0f;                     // sign of the K^-1 part of W_c
            const bool first = sigma == 0.f;
            const bool flip = !first && sg != sigma;
            const bool grow = !first && e_c > e_acc;
            if (flip || grow) {
                // the accumulators change sign and / or unit: a class of the other sign, or one whose M needs a larger bound
                const float fac = (flip ? -1.0f : 1.0f) * (grow ? ldexpf(1.0f, -2 * (e_c - e_acc)) : 1.0f);
#pragma unroll
                for (int n = 0; n < WREG; ++n) wr[n] *= fac;
#pragma unroll
                for (int n = 0; n < WLDS; ++n) wl[n * 64 + lane] = wl[n * 64 + lane] * fac;
            }
            e_acc = e_c;                                                    // (e2 was chosen >= e_acc)
            sigma = sg;
            const float unit = ldexpf(1.0f, 15 - e_acc);                    // accumulators hold sigma unit^2 W
#else
        // ---- phase 2 (fp32 matrix instruction): M = R^-T; M_ji (j > i) overwrites slot (i, j); M_kk sits in slot (k, k) ----
        f32x4 negI2;
#pragma unroll
        for (int q = 0; q < 4; ++q) negI2[q] = (g4 + q == c16) ? -TWOM15 : 0.0f;
#pragma unroll
        for (int j = 1; j < NT; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 nV = xty0(T.t[j][j], negI2);
#pragma unroll
            for (int k = 0; k < j; ++k) T.t[k][j] = join_h2(T.t[k][j]);
#pragma unroll
            for (int i = 0; i < j; i += 2) {
                if (i + 1 < j) {
                    f32x4 QA = xty0(T.t[i][j], T.t[i][i]);
                    f32x4 QB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = i + 1; k < j; ++k) xty2(T.t[k][j], T.t[i][k], QA, T.t[k][j], T.t[i + 1][k], QB);      // (k == i + 1: the diagonal slot)
                    f32x4 RA = {0.f, 0.f, 0.f, 0.f}, RB = {0.f, 0.f, 0.f, 0.f};
                    xty2(nV, QA, RA, nV, QB, RB);
                    T.t[i][j] = RA;
                    T.t[i + 1][j] = RB;
                } else {
                    f32x4 Q = xty0(T.t[i][j], T.t[i][i]);
#pragma unroll
                    for (int k = i + 1; k < j; ++k) Q = xty(T.t[k][j], T.t[i][k], Q);
                    T.t[i][j] = xty0(nV, Q);
                }
            }
        }
        DKT_PH(2);
        // ---- alpha = -(row N of M) / rho / sqrt(kappa) -> global, LDS ----
        const bool arow = (ln.g == (pN >> 2));
        const int qn = pN & 3;
        float asum = 0.f, aa = 0.f;
        {
            const brsrc ar = mk_rsrc(a.alpha + bc * N, (unsigned)(N * 4));
            const float asc = ldexpf(aug_unscale, -msc);
            float* ysf = reinterpret_cast<float*>(yst);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const f32x4 m = T.t[i][NT - 1];
                const float v = -(qn == 0 ? m[0] : qn == 1 ? m[1] : qn == 2 ? m[2] : m[3]) * asc;
                const bool ok = arow && ((i < NT - 1) || (c16 < pN));
                bstore1(ar, (fail_at != 0) ? qnan : v, ok ? (16 * i + c16) * 4 : OOB, 0);
                if (arow) ysf[16 * i + c16] = ok ? v : 0.f;
                asum += ok ? v : 0.f;
                aa += ok ? v * v : 0.f;
            }
        }
        DKT_PH(3);
        tq = threadIdx.x;
        DKT_OPAQUE_V(tq);
        c16 = tq & 15; g4 = (tq >> 2) & 12; lane = tq & 63;
        // ---- phase 3 on the f16 pipe, accumulated into the episode's W tiles ----
        const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
        const float wmag = 0.5f * fabsf(cw) * svc;                          // |coefficient| of (K^-1 - alpha alpha^T) in W
        const float gsc = __builtin_sqrtf(wmag) * ldexpf(1.0f, -msc);       // g_c: (g_c M)^T (g_c M) = wmag K^-1
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i) {                                      // the augmented row (-rho alpha_s^T, M_NN = 1) is zeroed
#pragma unroll
            for (int q = 0; q < 4; ++q) T.t[i][NT - 1][q] = (g4 + q == pN) ? 0.f : T.t[i][NT - 1][q];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i <= j; ++i) {
                const f32x4 v = T.t[i][j];
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            }
        }
        mx = wave_reduce_dpp<true>(mx) * gsc;
        mx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mx)));
        const bool live = want_grad && (fail_at == 0) && (wmag > 0.f) && (mx > 0.f) && (mx < 3.0e38f);     // a zero-weight class contributes nothing
        poison = poison || (fail_at != 0);
        float trk = 0.f;
        if (live) {
            const int e_c = (int)((__float_as_uint(mx) >> 23) & 0xffu) - 126;     // g_c max|M| = f 2^e_c, 0.5 <= f < 1
            const float sg = (cw > 0.f) ? -1.0f : 1.0f;                     // sign of the K^-1 part of W_c
            const bool first = sigma == 0.f;
            const bool flip = !first && sg != sigma;
            const bool grow = !first && e_c > e_acc;
            if (flip || grow) {
                // the accumulators change sign and / or unit: a class of the other sign, or one whose M needs a larger bound
                const float fac = (flip ? -1.0f : 1.0f) * (grow ? ldexpf(1.0f, -2 * (e_c - e_acc)) : 1.0f);
#pragma unroll
                for (int n = 0; n < WREG; ++n) wr[n] *= fac;
#pragma unroll
                for (int n = 0; n < WLDS; ++n) wl[n * 64 + lane] = wl[n * 64 + lane] * fac;
            }
            e_acc = first ? e_c : max(e_acc, e_c);
            sigma = sg;
            const float unit = ldexpf(1.0f, 15 - e_acc);                    // accumulators hold sigma unit^2 W
            const float qsc = gsc * unit;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int i = 0; i <= j; ++i) T.t[i][j] = split_h2(T.t[i][j], qsc);
            }
#endif
            // rank-one term: sigma unit^2 * (+0.5 cw sv) alpha alpha^T = -wmag unit^2 alpha_i alpha_j, into the chains' C operands
            const float* alf = reinterpret_cast<const float*>(yst);
            const float ca = -wmag * unit * unit;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                const float acol = alf[16 * j + c16] * ca;
#pragma unroll
                for (int i = 0; i <= j; i += 2) {
                    const bool pair = i + 1 <= j;
                    // C operands: the W accumulator + the rank-one term (off-diagonal); the rank-one term alone (diagonal: this class's trace is wanted)
                    auto cin = [&](const int ii) {
                        const f32x4 ar = yst[4 * ii + (g4 >> 2)];
                        const int n = tidx(ii, j);
                        f32x4 cacc = {0.f, 0.f, 0.f, 0.f};
                        if (ii < j) {
                            if (n < WLDS) cacc = wl[n * 64 + lane];
                            else cacc = wr[n - WLDS < 0 ? 0 : n - WLDS];
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) cacc[q] = fmaf(ar[q], acol, cacc[q]);
                        return cacc;
                    };
                    f32x4 accA = cin(i), accB = pair ? cin(i + 1) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = j; k < NT; ++k) {
                        const f32x4 Bm = T.t[j][k];                         // M_kj (k == j: the diagonal slot)
                        if (pair) xtyh2(T.t[i][k], Bm, accA, T.t[i + 1][k], Bm, accB);
                        else accA = xtyh(T.t[i][k], Bm, accA);
                    }
                    auto put = [&](const f32x4 acc, const int ii) {
                        const int n = tidx(ii, j);
                        if (ii < j) {
                            if (n < WLDS) wl[n * 64 + lane] = acc;
                            else wr[n - WLDS < 0 ? 0 : n - WLDS] = acc;
                        } else {
                            // diagonal tile: this class's sigma unit^2 wmag (K^-1 - alpha alpha^T)_jj; trace over the real rows, then into the accumulator
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const bool ok = (g4 + q == c16) && ((j < NT - 1) || (c16 < pN));
                                trk += ok ? acc[q] : 0.f;
                            }
                            if (n < WLDS) wl[n * 64 + lane] = wl[n * 64 + lane] + acc;
                            else wr[n - WLDS < 0 ? 0 : n - WLDS] += acc;
                        }
                    };
                    put(accA, i);
                    if (pair) put(accB, i + 1);
                }
            }
            trk *= 1.0f / (wmag * unit * unit);                             // tr (K^-1 - alpha alpha^T)
        }
        DKT_PH(4);
        lsum = wave_reduce_dpp<false>(lsum) + (float)(2 * msc * N);        // log2 det K = log2 det K_s + N log2 kappa
        asum = wave_reduce_dpp<false>(asum);
        aa = wave_reduce_dpp<false>(aa);
        trk = wave_reduce_dpp<false>(trk);
        if (want_grad && !live && fail_at == 0) {
            // zero-weight class: the trace is still wanted for the hyper-gradients -- tr K^-1 = |M|_F^2 (row N is zero by now)
            float fro = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
#pragma unroll
                for (int i = 0; i <= j; ++i) {
#if DKT_H2E_P2H
                    const f32x4 u = join_h2(T.t[i][j]);                     // the tiles hold g unit2 M as f16 splits
#else
                    const f32x4 u = T.t[i][j];
#endif
#pragma unroll
                    for (int q = 0; q < 4; ++q) fro += (j < NT - 1 || g4 + q < pN) ? u[q] * u[q] : 0.f;
                }
            }
#if DKT_H2E_P2H
            trk = wave_reduce_dpp<false>(fro) * ldexpf(1.0f, -2 * msc) / (gsc * gsc * unit2 * unit2) - aa;
#else
            trk = wave_reduce_dpp<false>(fro) * ldexpf(1.0f, -2 * msc) - aa;
#endif
        }
        if (lane == 0) {
            const bool ok = fail_at == 0;
            a.logp[bc] = ok ? (-0.5f * quad - 0.34657359027997264f * lsum - (float)N * DKT_HALF_LOG_2PI) : qnan;
            a.jitter_used[bc] = jit;
            a.info[bc] = (ok && kappa_high) ? -1 : fail_at;
            if (want_grad) {
                const float nz_eff = nzc + jit;
                a.dmean[bc] = ok ? asum : qnan;
                a.dnoise[bc] = ok ? -0.5f * trk : qnan;                                     // 0.5 (alpha.alpha - tr K^-1)
                a.dsv[bc] = ok ? 0.5f * ((quad - (float)N) + nz_eff * trk) / svc : qnan;
            }
        }
        DKT_PH(5);
        attempt = 0;
        ++c;
    }
    // ---- W[b] = sigma unit^-2 accumulators, stored once (a failed class poisons it) ----
    if (want_grad) {
        tq = threadIdx.x;
        DKT_OPAQUE_V(tq);
        c16 = tq & 15; g4 = (tq >> 2) & 12; lane = tq & 63; N = a.N;
        DKT_OPAQUE_S(N);
        const int pN = N - 16 * (NT - 1);
        const float fin = poison ? qnan : ((sigma == 0.f) ? 0.f : sigma * ldexpf(1.0f, -2 * (15 - e_acc)));
        const brsrc Wr = mk_rsrc(a.W + mat * N * N, (unsigned)(N * N * 4));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bool col_ok = (j < NT - 1) || (c16 < pN);
#pragma unroll
            for (int i = 0; i <= j; ++i) {
                const int n = tidx(i, j);
                f32x4 v;
                if (n < WLDS) v = wl[n * 64 + lane];
                else v = wr[n - WLDS < 0 ? 0 : n - WLDS];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = poison ? qnan : v[q] * fin;
                if (i < j) {
                    bstore4(Wr, v, col_ok ? ((16 * j + c16) * N + 16 * i + g4) * 4 : OOB, 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool row_ok = (i < NT - 1) || (g4 + q < pN);
                        bstore1(Wr, v[q], (row_ok && col_ok) ? ((16 * i + g4 + q) * N + 16 * j + c16) * 4 : OOB, 0);
                    }
                } else {
                    // diagonal tile: the upper triangle and its mirror (the h m + m h plane products are symmetric only up to rounding order)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool row_ok = (i < NT - 1) || (g4 + q < pN);
                        const bool keep = row_ok && col_ok && g4 + q <= c16;
                        bstore1(Wr, v[q], keep ? ((16 * i + g4 + q) * N + 16 * j + c16) * 4 : OOB, 0);
                        bstore1(Wr, v[q], (keep && g4 + q < c16) ? ((16 * j + c16) * N + 16 * i + g4 + q) * 4 : OOB, 0);
                    }
                }
            }
        }
    }
#ifdef DKT_MFMA_CLOCKS
    DKT_PH(6);
    if (lane == 0 && a.ws) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.ws) + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 7; ++i) o[i] = ph[i];
        o[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
#endif
}

// Batch size from which the wave-per-episode kernel serves the training call (below, one episode's classes run in parallel on five
// waves of the wave-per-matrix kernel).  DKT_MLL_H2E_MINB overrides (A/B runs, tests).
int g_h2e_minb = -1;
int h2e_min_batch() {
    if (g_h2e_minb < 0) { const char* e = getenv("DKT_MLL_H2E_MINB"); g_h2e_minb = e ? atoi(e) : 1024; }
    return g_h2e_minb;
}

template <int NT>
void launch_h2(const MllArgs& a, hipStream_t st) {
    const bool g = (a.flags & DKT_MLL_WANT_GRAD) != 0;
    if constexpr (NT <= 7) {
        if (a.flags & DKT_MLL_E_PER_CLASS) {               // per-class base matrices: always one wave per matrix, nothing shared
            hipLaunchKernelGGL((mll_h2e_kernel<NT>), dim3(a.B * a.C), dim3(64), 0, st, a);
            return;
        }
        if (g && a.B >= h2e_min_batch()) {
            hipLaunchKernelGGL((mll_h2e_kernel<NT>), dim3(a.B), dim3(64), 0, st, a);
            return;
        }
    }
    const bool epc = (a.flags & DKT_MLL_E_PER_CLASS) != 0;                  // (NT = 8 only: one wave per workgroup, one workgroup per matrix)
    const int rounds = (a.C + H2_MAX_WPG - 1) / H2_MAX_WPG;
    const int wpg = epc ? 1 : (a.C + rounds - 1) / rounds;
    constexpr int EPW = h2_epw<NT>();
    const dim3 grid(((epc ? a.B * a.C : a.B) + EPW - 1) / EPW), block(64 * wpg * EPW);
    if (g && wpg == 5) hipLaunchKernelGGL((mll_h2_kernel<NT, true, true>), grid, block, 0, st, a, wpg);
    else if (g) hipLaunchKernelGGL((mll_h2_kernel<NT, true, false>), grid, block, 0, st, a, wpg);
    else hipLaunchKernelGGL((mll_h2_kernel<NT, false, false>), grid, block, 0, st, a, wpg);
}

}  // namespace

void dkt_mll_h2_reload_env() { g_h2e_minb = -1; }        // dkt_reload_env(): tests / A-B tools flip DKT_MLL_H2E_MINB inside one process

// N + 1 <= 128 and no Cholesky output requested; false otherwise (the caller falls through to dkt_mll_mfma_launch).
bool dkt_mll_h2_launch(const MllArgs& a, hipStream_t st) {
    if (a.flags & DKT_MLL_WANT_CHOL) return false;
    const int nt = (a.N + 1 + 15) / 16;
    switch (nt) {
        case 1: launch_h2<1>(a, st); return true;
        case 2: launch_h2<2>(a, st); return true;
        case 3: launch_h2<3>(a, st); return true;
        case 4: launch_h2<4>(a, st); return true;
        case 5: launch_h2<5>(a, st); return true;
        case 6: launch_h2<6>(a, st); return true;
        case 7: launch_h2<7>(a, st); return true;
        case 8: launch_h2<8>(a, st); return true;
        default: return false;
    }
}
