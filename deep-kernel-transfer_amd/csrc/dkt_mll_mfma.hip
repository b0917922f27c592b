// dkt_mll_mfma.hip -- exact-GP marginal likelihood for N + 1 <= 128, ONE WAVE PER CLASS MATRIX, the whole
// factorisation / inversion / K^-1 product on v_mfma_f32_16x16x4_f32 with operands taken straight from accumulators.
//
// Replaces `-self.mll(output, targets)`, its autograd backward and the eval-mode mean cache
// (reference methods/DKT.py:161-163, 177, 187, 252-254, 265, 330; methods/DKT_regression.py:53-56, 92), i.e. GPyTorch's
// psd_safe_cholesky / inv_quad_logdet / cholesky_solve, for the C one-vs-rest models K_c = sv_c E + noise_c I of an episode.
//
// Workgroup = two episodes, one wave per class (C > 5: classes in rounds).  A wave keeps the upper block triangle of the
// (N+1)-augmented, 16-padded matrix
//        K' = [ K  r ; r^T 0 ] (+ identity on the padding),      r = y_c - m_c,
// as 16 x 16 tiles in the MFMA accumulator layout (lane (g, c), register q  <->  element [4g + q][c]; NT (NT+1) / 2 tiles,
// 4 VGPRs each).  The layout makes register q of a tile X a valid A operand (it is rows {q, q+4, q+8, q+12} of X,
// transposed) and register q of a tile Y a valid B operand (the same rows of Y), so
//        D += X^T Y   =   4 MFMAs, no LDS, no shuffles, no copies                                   (xty below)
// and every step of an UPPER (K' = R^T R) blocked algorithm is of that shape:
//   1. factorisation, block column k:  diagonal tile -> sweep (below) -> M_kk = R_kk^-T;   panel R_kj = (-V_kk)^T S_kj
//      (V_kk = M_kk^T through one X^T (-I) product);   trailing S_ij += R_ki^T R_kj.   Tiles hold S = -(Schur complement),
//      so the accumulate form of the MFMA is the update and no operand is ever negated.
//   2. inverse M = R^-T (lower), row j:  M_ji = (-V_jj)^T sum_{i<=k<j} R_kj^T M_ki, written over R_ij's slot.
//   3. K'^-1 = M^T M, in place:  P_ij = sum_{k>=j} M_ki^T M_kj.
// The augmented column does the vector work: column N of R is w = R^-T r, the Schur value met at pivot N is -|w|^2
// (the quadratic form), row N of M is -alpha^T, and with the sign of that row flipped in the A operand of step 3 the
// product is K^-1 - alpha alpha^T, i.e. -2 x the class's contribution to d logp / d K.  Pivot N and the padding pivots are
// forced to 1.
//
// Diagonal tile (the only VALU part, 16 pivots): the symmetric tile goes from the accumulator layout to a replicated
// column layout (every 16-lane row holds the whole tile, lane c = column c in 16 registers; v_permlane32_swap /
// v_permlane16_swap), where Gaussian elimination of [A | I] is one DPP-fused instruction per row and pivot:
//        a[i] += row_newbcast_p(a[i]) * (a[p] / -d),     b[i] += row_newbcast_p(a[i]) * (b[p] / -d)
// leaving d_p on the diagonal of a and, with rows scaled by 1/sqrt(d_p) as they become final, M_kk = R_kk^-T in b.
//
// W[b] = sum_c cls_weight_c sv_c 0.5 (alpha alpha^T - K_c^-1) is accumulated over the classes in LDS (tile layout, the
// waves take turns in a fixed order: deterministic) and stored ONCE, mirrored.  HBM traffic = E read + W written.
#include "dkt_mfma_tiles.h"

namespace {

using namespace dkt_mfma;

#ifdef DKT_MFMA_CLOCKS      // measurement build (tools/mll_phase_clocks.py): s_memtime stamps per wave into the workspace pointer
#define DKT_CLK(i) do { __builtin_amdgcn_sched_barrier(0); clk[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DKT_CLK(i) do { } while (0)
#endif
// Issue priority of a wave while it sweeps a diagonal tile (VALU work that competes for the shared fp32-MFMA / VALU pipe with the
// matrix instructions of the other waves of the SIMD): s_setprio.  0 = off.  Measured: -2 % (N = 105) / -3.5 % (N = 85) at 1; 3, or the
// whole factorisation at 1, are no different.
#ifndef DKT_MFMA_SWEEP_PRIO
#define DKT_MFMA_SWEEP_PRIO 1
#endif
#define DKT_SWEEP_PRIO(p) do { if (DKT_MFMA_SWEEP_PRIO) __builtin_amdgcn_s_setprio(p); } while (0)

constexpr int ntt(int nt) { return nt * (nt + 1) / 2; }
__host__ __device__ constexpr int tidx(int i, int j) { return j * (j + 1) / 2 + i; }      // i <= j (the order the tile loops enumerate)

constexpr int MFMA_MAX_WPG = 5;

template <int NT>
struct Tiles {
    f32x4 t[NT][NT];         // [i][j], i < j: off-diagonal slots;  [j][j]: diagonal slot
};

struct FormCtx {
    const f32x4* es;         // LDS: the episode's E tiles (raw, accumulator layout), shared by the waves of the episode
    const f32x4* ys;         // LDS: this wave's targets y_c, 16-byte groups
    int pN, c16, g4, lane;
    float nsv, dg, mc, rsc;  // -sv / kappa, -(noise + jitter) / kappa, mean, 1 / sqrt(kappa)
#ifdef DKT_MFMA_CLOCKS
    unsigned long long* clk2;
#endif
};

// Tile (I, J), I <= J, of S = -K' in the accumulator layout, from the staged E tile.  The last block column carries the augmented
// column -r (lanes c == pN), the last diagonal tile also its mirror row, a zero at the augmented pivot and -1 on the padding diagonal.
template <int NT, int I, int J>
__device__ __forceinline__ f32x4 form_tile(const FormCtx& f) {
    const int pN = f.pN, c16 = f.c16, g4 = f.g4;
    const f32x4 e = f.es[tidx(I, J) * 64 + f.lane];
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
                v = (g4 + q > pN) ? ((g4 + q == c16) ? -1.0f : 0.f) : v;        // padding: identity
                s[q] = v;
            }
        }
    }
    return s;
}

template <int NT, int I, int J>
__device__ __forceinline__ void form_row0(Tiles<NT>& T, const FormCtx& f) {      // tiles (0, J), J = 0 .. NT-1
    if constexpr (J < NT) {
        T.t[0][J] = form_tile<NT, 0, J>(f);
        form_row0<NT, I, J + 1>(T, f);
    }
}

// tile row 1 of block step 0's trailing update, the freshly formed tiles as C operands: S_1j = form(1, j) + R_01^T R_0j
template <int NT, int J>
__device__ __forceinline__ void form_trailing_row1(Tiles<NT>& T, const FormCtx& f) {
    if constexpr (J < NT) {
        T.t[1][J] = xty(T.t[0][1], T.t[0][J], form_tile<NT, 1, J>(f));
        form_trailing_row1<NT, J + 1>(T, f);
    }
}

// two independent X^T Y chains advanced alternately: the dependent-accumulate latency of one hides behind the other
__device__ __forceinline__ void xty2(const f32x4 xa, const f32x4 ya, f32x4& ca, const f32x4 xb, const f32x4 yb, f32x4& cb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ca = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[q], ya[q], ca, 0, 0, 0);
        cb = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[q], yb[q], cb, 0, 0, 0);
    }
}

// Trailing updates of block step K that are NOT needed by the next sweep (tile rows i >= K + 2), u = 0 .. n_pending - 1 in (i, j) order.
template <int NT, int K> constexpr int n_pending() { return (K >= 0 && NT - K - 2 > 0) ? (NT - K - 2) * (NT - K - 1) / 2 : 0; }
template <int NT, int K> constexpr int pend_i(int u) { int i = K + 2; while (u >= NT - i) { u -= NT - i; ++i; } return i; }
template <int NT, int K> constexpr int pend_j(int u) { int i = K + 2; while (u >= NT - i) { u -= NT - i; ++i; } return i + u; }

// The pending updates as a stream of single MFMAs, S = 0 .. n_pend_mfma - 1, ordered so that neighbours belong to different tiles
// (two tile updates advance alternately: the dependent-accumulate latency of one hides behind the other).
template <int NT, int K> constexpr int n_pend_mfma() { return 8 * ((n_pending<NT, K>() + 1) / 2); }

template <int NT, int K, int S0, int S1>
__device__ __forceinline__ void pend_mfma(Tiles<NT>& T, const FormCtx& f) {
    if constexpr (S0 < S1) {
        constexpr int u = 2 * (S0 / 8) + (S0 & 1), q = (S0 % 8) >> 1;
        if constexpr (u < n_pending<NT, K>()) {
            constexpr int i = pend_i<NT, K>(u), j = pend_j<NT, K>(u);
            if constexpr (K == 0 && q == 0) T.t[i][j] = form_tile<NT, i, j>(f);              // block step 0 consumes E as it goes
            T.t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(T.t[K][i][q], T.t[K][j][q], T.t[i][j], 0, 0, 0);
        }
        pend_mfma<NT, K, S0 + 1, S1>(T, f);
    }
}

// Slots of the sweep into which the pending MFMAs are dealt: one after every pivot's scalar chain and one after every piece of
// <= 5 row updates, i.e. about one MFMA (32 cycles of the matrix pipe) per 5-6 VALU instructions (~32 cycles of issue from one wave).
constexpr int slots_of_pivot(int p) { return 1 + (15 - p + 4) / 5; }
constexpr int slots_before(int p) { int n = 0; for (int i = 0; i < p; ++i) n += slots_of_pivot(i); return n; }
constexpr int SWEEP_SLOTS = slots_before(16);

template <int NT, int K, int SLOT>
__device__ __forceinline__ void run_slot(Tiles<NT>& T, const FormCtx& f) {
    constexpr int NM = n_pend_mfma<NT, K>();
    if constexpr (NM > 0) {
        pend_mfma<NT, K, SLOT * NM / SWEEP_SLOTS, (SLOT + 1) * NM / SWEEP_SLOTS>(T, f);
        __builtin_amdgcn_sched_barrier(0);                    // pin the MFMA between the VALU pieces
    }
}

template <int NT, int K, int P, int I0, int SLOT>
__device__ __forceinline__ void sweep_rows_slots(Tiles<NT>& T, const FormCtx& f, float (&x)[16], const float t) {
    if constexpr (I0 < 16) {
        constexpr int CNT = (16 - I0) < 5 ? (16 - I0) : 5;
        sweep_rows_piece<P, I0, CNT>(x, t);
        run_slot<NT, K, SLOT>(T, f);
        sweep_rows_slots<NT, K, P, I0 + CNT, SLOT + 1>(T, f, x, t);
    }
}

// The sweep of diagonal tile K + 1 with block step K's remaining trailing updates dealt over it: their MFMAs run on the matrix
// pipe underneath the sweep's VALU work of the same wave (K = -1: the very first tile, nothing pending).
template <int NT, int K, int P, bool LAST>
__device__ __forceinline__ void sweep_interleaved(Tiles<NT>& T, const FormCtx& f, float (&x)[16], float& dv, const Lane& ln, const int pn) {
    if constexpr (P < 16) {
        const float t = sweep_pivot_head<P, LAST>(x, dv, ln, pn);
#ifdef DKT_MFMA_CLOCKS
        if constexpr (K == -1) { __builtin_amdgcn_sched_barrier(0); f.clk2[P] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#endif
        run_slot<NT, K, slots_before(P)>(T, f);
        sweep_rows_slots<NT, K, P, P + 1, slots_before(P) + 1>(T, f, x, t);
        sweep_interleaved<NT, K, P + 1, LAST>(T, f, x, dv, ln, pn);
    }
}

template <bool CHOL>
struct P1Ctx {
    f32x4* myst;
    float* Lmat;             // CHOL: this matrix' L[N, N]
    float lsc;               // CHOL: sqrt(kappa), L = L_s sqrt(kappa)
    f32x4 negI;
    int lane, c16, pN, N;
    int fail_at;
    float lsum, quad;
#ifdef DKT_MFMA_CLOCKS
    unsigned long long* clk;
#endif
};

// Block step K of the factorisation; x / dv hold the swept diagonal tile K on entry.
template <int NT, int K, bool CHOL>
__device__ __forceinline__ void phase1_step(Tiles<NT>& T, const FormCtx& f, P1Ctx<CHOL>& c, float (&x)[16], float& dv, const Lane& ln) {
    if constexpr (K < NT) {
#ifdef DKT_MFMA_CLOCKS
        __builtin_amdgcn_sched_barrier(0);
        c.clk[13 + 2 * K] = __builtin_amdgcn_s_memtime();          // sweep K done
        __builtin_amdgcn_sched_barrier(0);
#endif
        const f32x4 M = sweep_end(x, ln);
        const bool valid = (K < NT - 1) || (c.c16 < c.pN);
        const unsigned long long badm = __ballot(valid && !(dv > 0.f)) & 0xffffull;
        const int first = (int)__builtin_ctzll(badm | 0x10000ull);
        c.fail_at = (c.fail_at == 0 && badm != 0) ? 16 * K + first + 1 : c.fail_at;
        c.lsum += (valid && ln.g0) ? __builtin_amdgcn_logf(dv) : 0.f;              // log2
        if constexpr (K == NT - 1) c.quad = -__int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv), c.pN));
        c.myst[K * 64 + c.lane] = M;
        if constexpr (CHOL) {
            // L[16K + c][16K + i] = R_KK[i][c] = -x[i] (i < c), 1 / M_KK[c][c] on the diagonal, zero above it
            if (ln.g0 && valid) {
                float* Lrow = c.Lmat + (size_t)(16 * K + c.c16) * c.N + 16 * K;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (16 * K + i < c.N) Lrow[i] = (i < c.c16) ? -x[i] * c.lsc : ((i == c.c16) ? c.lsc / x[i] : 0.f);
            }
        }
        if constexpr (K + 1 < NT) {
            const f32x4 nV = xty0(M, c.negI);                                        // M^T (-I) = -V_KK
#pragma unroll
            for (int j = K + 1; j < NT; ++j) T.t[K][j] = xty0(nV, T.t[K][j]);        // panel: R_Kj
            // tile row K + 1 first: the next sweep and the next panel need it
            if constexpr (K == 0) form_trailing_row1<NT, 1>(T, f);
            else {
#pragma unroll
                for (int j = K + 1; j < NT; ++j) T.t[K + 1][j] = xty(T.t[K][K + 1], T.t[K][j], T.t[K + 1][j]);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef DKT_MFMA_CLOCKS
            c.clk[14 + 2 * K] = __builtin_amdgcn_s_memtime();      // panel + tile row K + 1 issued
            __builtin_amdgcn_sched_barrier(0);
#endif
            DKT_SWEEP_PRIO(DKT_MFMA_SWEEP_PRIO);
            sweep_begin(T.t[K + 1][K + 1], x, dv);
            sweep_interleaved<NT, K, 0, K + 1 == NT - 1>(T, f, x, dv, ln, c.pN);
            DKT_SWEEP_PRIO(0);
            phase1_step<NT, K + 1, CHOL>(T, f, c, x, dv, ln);
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
        // last diagonal tile: the four columns may run past N (and past the end of E[b]): element-wise loads
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

// Episodes per workgroup.  The dispatcher deals the waves of a workgroup to the 4 SIMDs round-robin from SIMD 0, and at 3 waves per
// SIMD (168 VGPRs) a second 5-wave workgroup no longer fits on SIMD 0: measured one resident workgroup per CU.  Two episodes = 10
// waves per workgroup load the SIMDs (3, 3, 2, 2).  NT = 8 runs at 2 waves per SIMD (256 VGPRs): one episode per workgroup.
template <int NT> constexpr int mfma_epw() { return NT <= 7 ? 2 : 1; }

#define DKT_OPAQUE_V(x) asm volatile("" : "+v"(x))
#define DKT_OPAQUE_S(x) asm volatile("" : "+s"(x))

template <int NT, bool GRAD, bool CHOL, bool WPG5>
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64 * MFMA_MAX_WPG * mfma_epw<NT>()), amdgpu_waves_per_eu(NT <= 7 ? 3 : 2, NT <= 7 ? 3 : 2)))
void mll_mfma_kernel(MllArgs a, const int wpg) {
    constexpr int NTT = ntt(NT);
    constexpr int EPW = mfma_epw<NT>();
    // per episode: the E tiles (raw, shared by the class waves) -- and, once every wave is done with them, the exchange buffer of
    // the sum over the classes (5 waves x ceil(NTT / 5) tiles)
    __shared__ f32x4 stage_all[EPW][(GRAD && WPG5 ? 5 * ((NTT + 4) / 5) : NTT) * 64];
    __shared__ f32x4 mst[MFMA_MAX_WPG * EPW][NT * 64];           // per wave: the diagonal tiles M_kk
    __shared__ f32x4 yst[MFMA_MAX_WPG * EPW][NT * 4];            // per wave: the targets of its class

    const int tid = threadIdx.x;
    const int wall = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave in the workgroup
    const int epl = wall / wpg;                                  // episode within the workgroup
    const int w = wall - epl * wpg;                              // wave within the episode
    const int b = min(blockIdx.x * EPW + epl, a.B - 1);
    const bool ep_ok = blockIdx.x * EPW + epl < a.B;             // an odd tail: the surplus waves only keep the barriers company
    f32x4* const stage = stage_all[epl];
    f32x4* const myst = mst[wall];
    f32x4* const myys = yst[wall];
    const int C = a.C;
    const float qnan = __int_as_float(0x7fc00000);
    const int nrounds = (C + wpg - 1) / wpg;

#ifdef DKT_MFMA_CLOCKS
    unsigned long long clk[32] = {};
    unsigned long long clk2[16] = {};
#endif
    DKT_CLK(0);
    for (int round = 0; round < nrounds; ++round) {
        // lane coordinates and sizes made opaque per round: keeps the compiler from hoisting (and then spilling) every mask and
        // address of the round body into the kernel prologue
        int tq = tid;
        DKT_OPAQUE_V(tq);
        int c16 = tq & 15, g4 = (tq >> 2) & 12, lane = tq & 63, N = a.N;
        DKT_OPAQUE_S(N);
        Lane ln;
        ln.lane = lane; ln.g = g4 >> 2; ln.c = c16;
        ln.g0 = ln.g == 0; ln.g1 = ln.g == 1; ln.g2 = ln.g == 2;
        const int pN = N - 16 * (NT - 1);                        // local index of the augmented row / column in the last tile
        const int c = round * wpg + w;
        const bool active = ep_ok && c < C;
        // ---- stage E[b] (the waves of the episode share the loads) and this wave's targets ----
        if (ep_ok) {
            const brsrc Er = mk_rsrc(a.E + (size_t)b * N * N, (unsigned)(N * N * 4));
            stage_e<NT, 0, 0>(stage, Er, N, pN, c16, g4, lane, w, wpg);
        }
        if (active) {
            const brsrc yr = mk_rsrc(a.Y + (size_t)b * a.y_bstride + (size_t)c * N, (unsigned)(N * 4));
            float* ysf = reinterpret_cast<float*>(myys);
            if (lane < 16 * NT) ysf[lane] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, lane * 4, 0, 0));     // beyond N: 0
            if (NT > 4 && lane + 64 < 16 * NT) ysf[lane + 64] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(yr, (lane + 64) * 4, 0, 0));
        }
        __syncthreads();
        DKT_CLK(1);
        Tiles<NT> T;
        float coef = 0.f;
        if (active) {
            f32x4 negI;
#pragma unroll
            for (int q = 0; q < 4; ++q) negI[q] = (g4 + q == c16) ? -1.0f : 0.0f;
            const float svc = a.sv[c], mc = a.mean[c], nzc = a.noise[c];
            const size_t bc = (size_t)b * C + c;
            FormCtx f;
            f.es = stage; f.ys = myys;
#ifdef DKT_MFMA_CLOCKS
            f.clk2 = clk2;
#endif
            f.pN = pN; f.c16 = c16; f.g4 = g4; f.lane = lane; f.mc = mc;
            // Scale by kappa = 4^m >= max_i K_ii (exact in fp32): every pivot of K / kappa is <= 1, which the sweep's update at the
            // pivot lane relies on.  With r / 2^m in the augmented column, w and the quadratic form are unchanged;
            // alpha = alpha_s / 2^m, K^-1 - alpha alpha^T = (.)_s / kappa, log det K = log det K_s + N log kappa.
            float emax = 0.f;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f32x4 e = stage[tidx(j, j) * 64 + lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) emax = fmaxf(emax, (g4 + q == c16) ? e[q] : 0.f);
            }
            emax = wave_reduce_dpp<true>(emax);
            int fail_at = 0;
            float jit = 0.f, lsum = 0.f, quad = 0.f;
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
                f.nsv = -svc * ikap;
                f.dg = -(nzc + jit) * ikap;
                f.rsc = ldexpf(1.0f, -msc);
                form_row0<NT, 0, 0>(T, f);
                // ---- phase 1: factorisation.  The sweep of tile k + 1 runs ahead, interleaved with step k's remaining updates ----
                P1Ctx<CHOL> pc;
                pc.myst = myst; pc.Lmat = CHOL ? a.L + bc * N * N : nullptr; pc.lsc = ldexpf(1.0f, msc); pc.negI = negI;
                pc.lane = lane; pc.c16 = c16; pc.pN = pN; pc.N = N;
                pc.fail_at = 0; pc.lsum = 0.f; pc.quad = 0.f;
#ifdef DKT_MFMA_CLOCKS
                pc.clk = clk;
                clk[9] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, bits 0..31
                clk[12] = __builtin_amdgcn_s_memtime();
#endif
                {
                    float dv, x[16];
                    DKT_SWEEP_PRIO(DKT_MFMA_SWEEP_PRIO);
                    sweep_begin(T.t[0][0], x, dv);
                    sweep_interleaved<NT, -1, 0, NT == 1>(T, f, x, dv, ln, pN);
                    DKT_SWEEP_PRIO(0);
                    phase1_step<NT, 0, CHOL>(T, f, pc, x, dv, ln);
                }
                fail_at = pc.fail_at; lsum = pc.lsum; quad = pc.quad;
                if (fail_at == 0) break;
            }
            DKT_CLK(2);
            if constexpr (CHOL) {
                // off-diagonal tiles: L[16j + c][16k + 4g + q] = R_kj[4g+q][c]; the strictly upper tiles of L are zero
                const brsrc Lr = mk_rsrc(a.L + bc * N * N, (unsigned)(N * N * 4));
#pragma unroll
                for (int j = 1; j < NT; ++j) {
                    const bool row_ok = (j < NT - 1) || (c16 < pN);
#pragma unroll
                    for (int k = 0; k < j; ++k) {
                        bstore4(Lr, T.t[k][j] * ldexpf(1.0f, msc), row_ok ? ((16 * j + c16) * N + 16 * k + g4) * 4 : OOB, 0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool ok = (j < NT - 1) || (g4 + q < pN);
                            bstore1(Lr, 0.f, ok ? ((16 * k + c16) * N + 16 * j + g4 + q) * 4 : OOB, 0);
                        }
                    }
                }
            }
            // ---- phase 2: M = R^-T; M_ji (j > i) overwrites slot (i, j); two rows i at a time (independent MFMA chains) ----
            // (lane coordinates opaque again: what the later phases derive from them is recomputed here instead of being kept
            // alive -- spilled -- across the factorisation)
            tq = tid;
            DKT_OPAQUE_V(tq);
            c16 = tq & 15; g4 = (tq >> 2) & 12; lane = tq & 63;
            f32x4 negI2;
#pragma unroll
            for (int q = 0; q < 4; ++q) negI2[q] = (g4 + q == c16) ? -1.0f : 0.0f;
#pragma unroll
            for (int j = 1; j < NT; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 nV = xty0(myst[j * 64 + lane], negI2);
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
            // ---- alpha = -(row N of M) ----
            const bool arow = (ln.g == (pN >> 2));
            const int qn = pN & 3;
            float asum = 0.f;
            {
                const brsrc ar = mk_rsrc(a.alpha + bc * N, (unsigned)(N * 4));
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const f32x4 m = (i < NT - 1) ? T.t[i][NT - 1] : myst[(NT - 1) * 64 + lane];
                    const float v = -(qn == 0 ? m[0] : qn == 1 ? m[1] : qn == 2 ? m[2] : m[3]) * f.rsc;
                    const bool ok = arow && ((i < NT - 1) || (c16 < pN));
                    bstore1(ar, (fail_at != 0) ? qnan : v, ok ? (16 * i + c16) * 4 : OOB, 0);
                    asum += ok ? v : 0.f;
                }
            }
            float trpp = 0.f;
            DKT_CLK(4);
            if constexpr (GRAD) {
                // ---- phase 3: P''_ij = sum_{k >= j} flip(M_ki)^T M_kj, in place (flip: the sign of row N) ----
                f32x4 sgn;
#pragma unroll
                for (int q = 0; q < 4; ++q) sgn[q] = (g4 + q == pN) ? -1.0f : 1.0f;
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
                            f32x4 A0 = (k == i) ? myst[k * 64 + lane] : T.t[i][k];
                            if (k == NT - 1) A0 = A0 * sgn;
                            if (pair) {
                                f32x4 A1 = (k == i + 1) ? myst[k * 64 + lane] : T.t[i + 1][k];
                                if (k == NT - 1) A1 = A1 * sgn;
                                xty2(A0, Bm, accA, A1, Bm, accB);
                            } else {
                                accA = xty(A0, Bm, accA);
                            }
                        }
                        auto put = [&](const f32x4 acc, const int ii) {
                            if (ii < j) {
                                T.t[ii][j] = acc;
                            } else {
                                // diagonal tile: M_jj is dead now, P''_jj takes its place in LDS; trace over the real rows
                                myst[j * 64 + lane] = acc;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const bool ok = (g4 + q == c16) && ((j < NT - 1) || (c16 < pN));
                                    trpp += ok ? acc[q] : 0.f;
                                }
                            }
                        };
                        put(accA, i);
                        if (pair) put(accB, i + 1);
                    }
                }
            }
            DKT_CLK(5);
            lsum = wave_reduce_dpp<false>(lsum) + (float)(2 * msc * N);            // log2 det K = log2 det K_s + N log2 kappa
            trpp *= ldexpf(1.0f, -2 * msc);
            asum = wave_reduce_dpp<false>(asum);
            trpp = wave_reduce_dpp<false>(trpp);
            if (lane == 0) {
                const bool ok = fail_at == 0;
                a.logp[bc] = ok ? (-0.5f * quad - 0.34657359027997264f * lsum - (float)N * DKT_HALF_LOG_2PI) : qnan;
                a.jitter_used[bc] = jit;
                a.info[bc] = fail_at;
                if constexpr (GRAD) {
                    const float nz_eff = a.noise[c] + jit;                          // (re-read: cheaper than keeping them across the phases)
                    a.dmean[bc] = ok ? asum : qnan;
                    a.dnoise[bc] = ok ? -0.5f * trpp : qnan;                                    // 0.5 (alpha.alpha - tr K^-1)
                    a.dsv[bc] = ok ? 0.5f * ((quad - (float)N) + nz_eff * trpp) / a.sv[c] : qnan;
                }
            }
            if constexpr (CHOL) {
                if (fail_at != 0) {
                    float* Lb = a.L + bc * N * N;
                    for (int idx = lane; idx < N * N; idx += 64) Lb[idx] = qnan;
                }
            }
            const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
            coef = (fail_at == 0) ? -0.5f * cw * a.sv[c] * ldexpf(1.0f, -2 * msc) : qnan;   // W_c = coef P''_s   (a failed class poisons W[b])
        }
        DKT_CLK(6);
        if constexpr (GRAD) {
            // ---- W[b] = sum over the classes of coef_c P''_c.  The staged E is dead once every wave is past its factorisation
            // (the barrier); its LDS becomes the exchange buffer. ----
            __syncthreads();
            tq = tid;
            DKT_OPAQUE_V(tq);
            c16 = tq & 15; g4 = (tq >> 2) & 12; lane = tq & 63;
            DKT_OPAQUE_S(N);
            const int pNs = N - 16 * (NT - 1);
            const brsrc Wr = mk_rsrc(a.W + (size_t)b * N * N, (unsigned)(N * N * 4));
            // tile n -> W[b]: element [4g+q][c] of tile (i, j) and its mirror; later rounds add to what this very wave stored before
            auto store_tile = [&](f32x4 v, const int i, const int j) {
                const bool col_ok = (j < NT - 1) || (c16 < pNs);
                const int vo_m = col_ok ? ((16 * j + c16) * N + 16 * i + g4) * 4 : OOB;       // i < j <= NT-1: real columns
                int vo_d[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool row_ok = (i < NT - 1) || (g4 + q < pNs);
                    vo_d[q] = (row_ok && col_ok) ? ((16 * i + g4 + q) * N + 16 * j + c16) * 4 : OOB;
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
            };
            if constexpr (WPG5) {
                // Five chunks (tile n -> chunk n % 5).  Per chunk every wave drops its <= 6 scaled tiles into its own slots of the
                // exchange buffer (compile-time register tiles, no divergent access to them); after the barrier wave w sums the
                // five copies of tile 5 w + chunk (wave 0 also of tile 25 + chunk) in class order -- deterministic -- and stores it.
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
                                if (active) v = ((i < j) ? T.t[i][j] : myst[j * 64 + lane]) * coef;
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
                                f32x4 v = ((i < j) ? T.t[i][j] : myst[j * 64 + lane]) * coef;
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
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.ws) + ((size_t)blockIdx.x * (MFMA_MAX_WPG * EPW) + wall) * 48;
        for (int i = 0; i < 32; ++i) o[i] = clk[i];
        for (int i = 0; i < 16; ++i) o[32 + i] = clk2[i];
    }
#endif
}

template <int NT>
void launch_mfma(const MllArgs& a, hipStream_t st) {
    const bool g = (a.flags & DKT_MLL_WANT_GRAD) != 0, c = (a.flags & DKT_MLL_WANT_CHOL) != 0;
    const int rounds = (a.C + MFMA_MAX_WPG - 1) / MFMA_MAX_WPG;
    const int wpg = (a.C + rounds - 1) / rounds;
    constexpr int EPW = mfma_epw<NT>();
    const int epw = EPW;
    const dim3 grid((a.B + epw - 1) / epw), block(64 * wpg * epw);
    if (g && c && wpg == 5) { hipLaunchKernelGGL((mll_mfma_kernel<NT, true, true, true>), grid, block, 0, st, a, wpg); return; }
    if (g && c) { hipLaunchKernelGGL((mll_mfma_kernel<NT, true, true, false>), grid, block, 0, st, a, wpg); return; }
    if (c) { hipLaunchKernelGGL((mll_mfma_kernel<NT, false, true, false>), grid, block, 0, st, a, wpg); return; }
#ifdef DKT_TWINS         // without the Cholesky output this kernel is the exact-fp32 twin of dkt_mll_h2.hip (DKT_MLL_FORCE_F32MFMA)
    if (g && wpg == 5) hipLaunchKernelGGL((mll_mfma_kernel<NT, true, false, true>), grid, block, 0, st, a, wpg);
    else if (g) hipLaunchKernelGGL((mll_mfma_kernel<NT, true, false, false>), grid, block, 0, st, a, wpg);
    else hipLaunchKernelGGL((mll_mfma_kernel<NT, false, false, false>), grid, block, 0, st, a, wpg);
#endif
}

}  // namespace

// The product library serves DKT_MLL_WANT_CHOL here for N + 1 <= 32 (the regression head's conditioning sets, DKT_regression.py:84-93: 5 .. 19 rows; larger
// requests take the generic kernel) and nothing else; the twins library (-DDKT_TWINS) has every size with and without the Cholesky output -- the exact-fp32
// twin of the f16-split kernels.  Returns false when the call is not served.
bool dkt_mll_mfma_launch(const MllArgs& a, hipStream_t st) {
    const int nt = (a.N + 1 + 15) / 16;
#ifndef DKT_TWINS
    if (!(a.flags & DKT_MLL_WANT_CHOL) || nt > 2) return false;
#endif
    switch (nt) {
        case 1: launch_mfma<1>(a, st); return true;
        case 2: launch_mfma<2>(a, st); return true;
#ifdef DKT_TWINS
        case 3: launch_mfma<3>(a, st); return true;
        case 4: launch_mfma<4>(a, st); return true;
        case 5: launch_mfma<5>(a, st); return true;
        case 6: launch_mfma<6>(a, st); return true;
        case 7: launch_mfma<7>(a, st); return true;
        case 8: launch_mfma<8>(a, st); return true;
#endif
        default: return false;
    }
}
