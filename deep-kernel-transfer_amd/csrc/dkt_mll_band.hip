// dkt_mll_band.hip -- exact-GP marginal likelihood of the C one-vs-rest models of an episode that SHARE their base matrix (linear / cossim / bncossim,
// 128 <= N <= 432, 12 <= C <= 32, >= 192 episodes per call: the 20-way shapes of train.py:132-133) through ONE orthogonal reduction per episode instead of C factorisations.
//
// Replaces the same reference lines as dkt_mll_tiled.hip (methods/DKT.py:161-163 at C = 20: GPyTorch's psd_safe_cholesky / inv_quad_logdet / cholesky_solve
// and their autograd backward).  DKT.py:148-149 hands every class model the same z_train and :346-347 freezes the noise, so the class matrices are shifts of
// one matrix:   K_c = sv_c (E + mu_c I),  mu_c = noise_c / sv_c.   With B = Q^T E Q block tridiagonal (16 x 16 blocks; Q = H_0 H_1 ..., H_k = I - V_k T_k V_k^T
// the compact-WY form of the 16 Householder reflectors of panel k):
//     log det K_c = N log sv_c + sum log pivots of the block LDL^T of B + mu_c I                       (a non-positive pivot = attempt 0 of the ladder failed)
//     alpha_c     = Q a_c / sv_c,   a_c = (B + mu_c)^-1 Q^T r_c
//     W           = Q [ sum_c 0.5 cw_c (a_c a_c^T / sv_c - (B + mu_c)^-1) ] Q^T                        (ONE similarity transform per episode)
//     tr(K_c^-1), tr(K_c^-1 E), alpha^T E alpha from tr (B + mu_c)^-1, a.a, a.u                        (hyper-gradients: scalars)
// tools/band_mll_model.py is the executable statement of the algorithm (tests/test_band_model.py holds it to the float64 oracle; float32: 1e-6 on the
// log-likelihood, 7e-6 on W at N = 420).  Flops per episode: ~6.5 N^3 instead of C 4/3 N^3 = 26.7 N^3; memory: one N x N tile array per EPISODE instead of one
// per class matrix.
//
// Everything is 16 x 16 tiles of 1 KB in the MFMA accumulator layout (lane (g, c) register q <-> element [4g + q][c]), every product is D += X^T Y =
// 4 x v_mfma_f32_16x16x4_f32 (dkt_mfma_tiles.h).  The forward reduction, the class chains and the column chains are EXACT fp32 (their rounding is amplified by
// cond(K)); only the back transform -- orthogonal similarities of M, every operand bounded a priori -- runs as scaled 2-way f16 splits (dkt_h2_tiles.h).
// A / B / M live as the LOWER block triangle of the episode (tile (i, j), i >= j, at i (i + 1) / 2 + j).  Kernels (workspace per episode in `BandGeo`):
//   band_init_kernel        E[b] -> the tile array, U = [y_c - m_c] as tiles
//   band_sym_kernel<false>  (workgroup = episode, 2 per CU)  panels k = 0 .. NT-3: update block column k with the finished panel, Householder QR of it (rows over
//                           threads, ONE exchange per column), then ONE fused pass over the stored tiles: A_ij -= V_i Y_j^T + Y_i V_j^T and the next panel's
//                           X' = A V' from the same registers (per-wave partial accumulators for every tile column, a wave-private LDS transpose for the mirrored
//                           contribution); T' by the larft recurrence from V'^T V', Y' = X' Th^T - 0.5 V' (Th S Th^T), U <- H'^T U
//   band_class_kernel       (wave = class matrix)  block LDL^T chain of B + mu_c I on the diagonal-tile sweep of dkt_mfma_tiles.h: pivots, P_j^-1, G_j = S_j P_j^-1,
//                           forward / backward substitution for a_c, Z_jj = diagonal blocks of (B + mu_c)^-1, tr Z, a.a, the a-priori condition bound
//   band_chain_kernel       (workgroup = episode, 8 waves x 4 block columns)  Z_ji = -G_j^T Z_{j+1,i} upwards from the diagonal, two classes per step, accumulated
//                           over the classes in registers, the classes' G tiles DMA-staged through LDS, + the rank-C term
//   band_sym_kernel<true>   M <- H_k M H_k^T for k = NT-3 .. 0 (the same fused pass with Th = T, f16 splits; its tile stream carries the non-temporal hint: at 17 GB per
//                           1024 episodes of 420 rows this kernel runs at the memory system's rate), a <- H_k a
//   band_finish_kernel      (workgroup = 16-row stripe of an episode)  W[b], alpha[b], and the quadratic form / hyper-gradients from the residual rho = r - K alpha against
//                           the ORIGINAL E (the reduction's backward error, a few eps |E|, would otherwise cost 3e-5 ... 9e-5 on r^T K^-1 r for class-correlated features);
//                           band_reduce_kernel sums the stripes in a fixed order
// Attempt 0 only (no jitter): an episode with a failed class -- or with a class whose a-priori bound 1 + sv trace(E) / noise exceeds BAND_KAPPA_MAX -- is redone,
// jitter ladder and all, by the generic kernel's fix-up launch, as in the tile-array path.  Measurements and everything that was tried: docs/MEASUREMENTS.md R6.
#include "dkt_h2_tiles.h"
#include <type_traits>

namespace {

using namespace dkt_mfma;

constexpr int BAND_MAXNT = 27;             // N <= 432 (the accumulators of the pass / chain kernels are sized for it)
constexpr int LDP = 20;                    // row stride (floats) of the panel-shaped LDS arrays: 16-byte aligned rows
constexpr int BAND_CHUNK = 1024;           // episodes per pass over the workspace
constexpr float BAND_KAPPA_MAX = 2.0e4f;   // a-priori condition bound above which an episode goes to the generic kernel (3 eps kappa < 4e-3 on alpha / W below it: measured 1e-5 .. 1e-4 at kappa ~ 1e2 .. 1e3)
static_assert(BAND_MAXNT % 9 == 0, "the partial-sum rounds walk the tiles nine at a time");           // episodes per pass over the workspace

struct BandGeo {
    int N, NT, C, CP;                      // CP = class-column tiles of U / A
    int oA, oV, oT, oU, oAm, oAmT, oG, oZd, oPi, oZv, oSc, oVsp;      // offsets (floats) into the episode's workspace
    int ep_floats;
};

struct BandArgs {
    MllArgs a;
    BandGeo g;
    float* ws;
    int b0, bcnt;
    int grad;
};

__host__ __device__ inline int band_voff(const int NT, const int k) { return k * (NT - 1) - (k * (k - 1)) / 2; }      // first tile of panel k's V (NT - k - 1 tiles)

BandGeo band_geo(int N, int C) {
    BandGeo g;
    g.N = N; g.NT = (N + 15) / 16; g.C = C; g.CP = (C + 15) / 16;
    int o = 0;
    auto take = [&](int tiles) { const int r = o; o += tiles * 256; return r; };
    g.oA = take(g.NT * (g.NT + 1) / 2);           // the lower block triangle, tile (i, j) at i (i + 1) / 2 + j
    g.oV = take(g.NT * (g.NT - 1) / 2);
    g.oT = take(g.NT);
    g.oU = take(g.NT * g.CP);
    g.oAm = take(g.NT * g.CP);
    g.oAmT = take(g.NT * g.CP);
    g.oG = take(C * g.NT);
    g.oZd = take(C * g.NT);
    g.oPi = take(C * g.NT);
    g.oZv = take((C * g.NT + 15) / 16);     // 16 floats per (class, block)
    g.oSc = take(1);                        // per class: a.a (the bound of |M| the back pass scales its f16 splits with)
    g.oVsp = take(g.NT);                    // the back pass: the next panel's V tiles, split
    g.ep_floats = o;
    return g;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 neg4(const f32x4 v) { return (f32x4){-v[0], -v[1], -v[2], -v[3]}; }
// the negative of an operand tile: fp32 elements, or (SPLIT) the four packed f16 pairs of a split tile -- the sign bits of both halves
template <bool SPLIT>
__device__ __forceinline__ f32x4 negop(const f32x4 v) {
    if constexpr (!SPLIT) return neg4(v);
    else return (f32x4){__uint_as_float(__float_as_uint(v[0]) ^ 0x80008000u), __uint_as_float(__float_as_uint(v[1]) ^ 0x80008000u),
                        __uint_as_float(__float_as_uint(v[2]) ^ 0x80008000u), __uint_as_float(__float_as_uint(v[3]) ^ 0x80008000u)};
}
constexpr f32x4 ZERO4 = {0.f, 0.f, 0.f, 0.f};

// ------------------------------------------------------------------------------------------------------------------------------------------------
// E[b] -> A tiles (lower block triangle), U tiles
// ------------------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_init_kernel(BandArgs t) {
    const BandGeo& G = t.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c16 = lane & 15, g4 = (lane >> 2) & 12;
    const int NT = G.NT, N = G.N, C = G.C;
    const int slot = blockIdx.x * 4 + wave, bl = blockIdx.y;
    float* ep = t.ws + (size_t)bl * G.ep_floats;
    const int b = t.b0 + bl;
    const int ntt = NT * (NT + 1) / 2;
    if (slot < ntt) {
        int i = 0, rem = slot;
        while (rem > i) { rem -= i + 1; ++i; }
        const int j = rem;                                  // i >= j
        // element [4g + q][c] = E[16i + 4g + q][16j + c] = E[16j + c][16i + 4g + q] (symmetric): one 16-byte load per lane
        const brsrc Er = mk_rsrc(t.a.E + (size_t)b * N * N, (unsigned)((size_t)N * N * 4));
        const int row = 16 * j + c16, col = 16 * i + g4;
        f32x4 e;
        if (col + 3 < N) {
            e = bload4(Er, row < N ? (row * N + col) * 4 : OOB, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                e[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Er, (row < N && col + q < N) ? (row * N + col + q) * 4 : OOB, 0, 0));
        }
        st4(ep + G.oA + (size_t)slot * 256 + lane * 4, e);
    } else if (slot < ntt + NT * G.CP) {
        const int u = slot - ntt, i = u / G.CP, p = u - i * G.CP;
        const int cls = 16 * p + c16;
        const float* Y = t.a.Y + (size_t)b * t.a.y_bstride;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * i + g4 + q;
            float x = 0.f;
            if (r < N) {
                if (cls < C) x = Y[(size_t)cls * N + r] - t.a.mean[cls];
            }
            v[q] = x;
        }
        st4(ep + G.oU + (size_t)u * 256 + lane * 4, v);
    } else if (slot < ntt + 3 * NT * G.CP) {
        // A / A^T tiles: the class kernel writes the columns of its classes only; the others (the padding) must read as zero
        st4(ep + G.oAm + (size_t)(slot - ntt - NT * G.CP) * 256 + lane * 4, ZERO4);       // (oAmT follows oAm)
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// The two-sided kernel.  BACK = false: reduction of A to block-tridiagonal form, panels k = 0 .. NT-3 in order; BACK = true: M <- Q M Q^T, panels in reverse.
// A lives as its LOWER block triangle (tile (i, j), i >= j, slot i (i + 1) / 2 + j: tile rows contiguous).  Per panel ONE pass over the stored tiles does both
// the update with the panel just finished (V, Y in LDS) and the product X' = A V' with the next panel (V' known: the forward pass factors block column k + 1
// right after updating it; the back pass reads V' from the workspace): every stored tile is read once and written once per panel.
//   wave = tile rows (snake order).  Tile (i, j):  A_ij -= V_i Y_j^T + Y_i V_j^T;   Xt'_j += V'_i^T A_ij  (per-wave partial accumulators for every j, in registers);
//   Xt'_i += V'_j^T A_ij^T for j < i (the tile transposed through 1 KB of wave-private LDS).  The partials of the four waves meet in LDS in a fixed order.
// ------------------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lslot(const int i, const int j) { return (i * (i + 1)) / 2 + j; }          // i >= j

#ifdef DKT_BAND_CLOCKS
#define BCLK(i) do { __builtin_amdgcn_sched_barrier(0); if (tid == 0) clkl[i] += (float)(__builtin_amdgcn_s_memtime() - tlast); tlast = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define BCLK(i) do { } while (0)
#endif

// Sums of 16 per-lane values over the 64 lanes of a wave as a reduce-scatter: every stage halves the values a lane carries (it keeps the half its lane bit selects
// and adds the partner's copy of that half).  On return lane l holds the total of value index 8 b5 + 4 b4 + 2 b3 + b2 (b_k = bit k of l).  All on the VALU:
// lane bits 5 / 4 with v_permlane32_swap / v_permlane16_swap on PAIRS of values (one swap + one add per output), bit 3 with row_ror:8, bit 2 with
// row_half_mirror (partner l ^ 7: the same bits 3..5, and bits 0 / 1 are summed afterwards anyway), bits 1 / 0 with quad_perm.  (As __shfl_xor it was 17
// ds_bpermute_b32 -- six dependent trips through the LDS crossbar per QR column, and six lane-address registers alive across the whole kernel.)
__device__ __forceinline__ float wave_sum16(const float (&v)[16], const int lane) {
    const bool h3 = (lane & 8) != 0, h2 = (lane & 4) != 0;
    float a[8], b[4], c[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[8 + i]), false, false);
        a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[4 + i]), false, false);
        b[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#define DKT_BDPP(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false))
#pragma unroll
    for (int i = 0; i < 2; ++i) c[i] = (h3 ? b[2 + i] : b[i]) + DKT_BDPP(h3 ? b[i] : b[2 + i], 0x128);     // row_ror:8 = lane ^ 8
    float d = (h2 ? c[1] : c[0]) + DKT_BDPP(h2 ? c[0] : c[1], 0x141);                                     // row_half_mirror = lane ^ 7
    d += DKT_BDPP(d, 0x4E);                                                                               // quad_perm [2, 3, 0, 1]
    d += DKT_BDPP(d, 0xB1);                                                                               // quad_perm [1, 0, 3, 2]
#undef DKT_BDPP
    return d;
}

struct SymLds {
    float* Vs;       // [NP - 16][LDP]  V of the finished panel, row-major, panel-local rows (row 0 = global row 16 r_prev)
    float* Ys;       // [NP][LDP]       Y of the finished panel / X' of the next one, rows = global rows
    float* Ts;       // [16][LDP]       T row-major
    float* Part;     // [4][256]        per-wave partial tile (S)
    float* Gs;       // [256]           V'^T V'
    float* Wus;      // [2][256]        V'^T U
    float* red;      // [2][4][17]      QR exchange
    float* rowj;     // [2][16]
    float* taus;     // [16]            tau of the panel being factored
    float* Tsc;      // [4][16][17]     wave-private transposition scratch
};
constexpr int TSC_LD = 17;
// cache policy of the BACK pass' tile stream (each stored tile is read once and written once per panel; 2 = non-temporal, 0 = default).  The forward kernel keeps the
// default: there the tiles a pass writes are read again right away (column update, panel load), and the hint cost 0.3 ms per 1024 episodes (measured)
#ifndef DKT_BAND_BACK_LD_POL
#define DKT_BAND_BACK_LD_POL 2
#endif
#ifndef DKT_BAND_BACK_ST_POL
#define DKT_BAND_BACK_ST_POL 2
#endif

__host__ __device__ inline int sym_lds_floats(const int NT) {
    const int NP = 16 * NT;
    return (NP - 16) * LDP + NP * LDP + 16 * LDP + 4 * 256 + 256 + 2 * 256 + 2 * 4 * 17 + 2 * 16 + 16 + 4 * 16 * TSC_LD + 16;          // (TSC: 272 floats per wave >= the 256 of the sink above; + the phase clocks of the measurement build)
}

// the wave that owns tile row i: rows dealt in snake order from the longest (i = NT - 1) down, so that the four waves get about the same number of tiles
__device__ __forceinline__ int row_owner(const int NT, const int i) {
    const int idx = NT - 1 - i;
    return (idx & 4) ? 3 - (idx & 3) : (idx & 3);
}

template <bool BACK>
__global__ __launch_bounds__(256, 2) void band_sym_kernel(BandArgs t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const BandGeo& G = t.g;
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g4 = (lane >> 2) & 12;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = G.NT, NP = 16 * NT, N = G.N, CP = G.CP;
    const int bl = blockIdx.x;
    float* ep = t.ws + (size_t)bl * G.ep_floats;
    float* At = ep + G.oA;
    float* Vg = ep + G.oV;
    float* Tg = ep + G.oT;
    float* Ut = ep + (BACK ? G.oAm : G.oU);
    const brsrc Ar = mk_rsrc(At, (unsigned)((size_t)NT * (NT + 1) / 2 * 1024));
    const brsrc Vr = mk_rsrc(Vg, (unsigned)((size_t)NT * (NT - 1) / 2 * 1024));
    SymLds L;
    L.Vs = smem;
    L.Ys = L.Vs + (NP - 16) * LDP;
    L.Ts = L.Ys + NP * LDP;
    L.Part = L.Ts + 16 * LDP;
    L.Gs = L.Part + 4 * 256;
    L.Wus = L.Gs + 256;
    L.red = L.Wus + 2 * 256;
    L.rowj = L.red + 2 * 4 * 17;
    L.taus = L.rowj + 2 * 16;
    L.Tsc = L.taus + 16 + wave * 16 * TSC_LD;
#ifdef DKT_BAND_CLOCKS
    float* const clkl = L.Tsc - wave * 16 * TSC_LD + 4 * 16 * TSC_LD;        // 12 floats of LDS behind the scratch (thread 0 only)
    if (tid == 0) for (int i = 0; i < 12; ++i) clkl[i] = 0.f;
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    const bool want_m = !BACK || t.grad;          // the back pass without gradients only carries the class vectors (alpha)
    const int npan = NT - 2;
    const int lane16 = lane * 16;
    // The back pass runs its tile products as scaled 2-way f16 splits (3 x v_mfma_f32_16x16x16_f16 = 24 cycles instead of 4 x 32; dkt_h2_tiles.h).  It transforms
    // M by ORTHOGONAL similarities, so every entry it ever holds is bounded by |M|_2 <= sum_c 0.5 |cw_c| (sv_c / noise_c + a_c.a_c / sv_c) -- known before the
    // pass -- the Householder vectors by 1, and Y by its own maximum, taken when it is formed: the splits keep 22 bits below those bounds, and a rounding here is
    // NOT amplified by the conditioning of K (the forward reduction, where it would be, stays exact fp32).
    constexpr float sV = 32768.0f, isV = 1.0f / 32768.0f;
    float sA = 1.0f, isA = 1.0f, sY = 1.0f, isY = 1.0f;
    [[maybe_unused]] const brsrc Vsr = mk_rsrc(ep + G.oVsp, (unsigned)(NT * 1024));
    if constexpr (BACK) {
        if (want_m) {
            float bm = 0.f;
            if (tid < G.C) {
                const float cwc = t.a.cls_weight ? t.a.cls_weight[tid] : 1.0f, svc = t.a.sv[tid];
                bm = 0.5f * fabsf(cwc) * (svc / t.a.noise[tid] + ep[G.oSc + tid] / svc);
            }
            if (wave == 0) {
                bm = wave_reduce_dpp<false>(bm);
                if (lane == 0) L.red[0] = bm;
            }
            __syncthreads();
            bm = L.red[0];
            if (bm > 0.f && bm < 1e30f) sA = scale_for(bm, isA);
            {                                                                               // the first panel's tiles, split
                const int k0 = NT - 3, vb0 = band_voff(NT, k0);
                for (int i2 = wave; i2 < NT - k0 - 1; i2 += 4)
                    bstore4(Vsr, split_h2(ld4(Vg + (size_t)(vb0 + i2) * 256 + lane * 4), sV), lane16, i2 * 1024);
            }
            __syncthreads();
        }
    }

    for (int it = 0; it <= npan; ++it) {
        const bool have_prev = it > 0, have_next = it < npan;
        const int kp = BACK ? NT - 2 - it : it - 1, kn = BACK ? NT - 3 - it : it;       // finished panel / next panel
        const int r_prev = kp + 1, r_next = kn + 1;                                      // first tile row of V / of V'
        const int mtn = NT - r_next;                                                     // tiles of V'
        const int vb_next = band_voff(NT, kn);
        if constexpr (!BACK) {
            if (have_next) {
                // ---- block column kn: update with the finished panel, then its Householder QR (thread t owns the panel rows t and t + 256) ----
                if (have_prev) {
                    const f32x4 ytj = ld4(L.Ys + (16 * kn + c16) * LDP + g4), vtj = ld4(L.Vs + (c16) * LDP + g4);      // j = kn = r_prev: V's first tile
                    for (int i = kn + wave; i < NT; i += 4) {
                        const f32x4 nvt = neg4(ld4(L.Vs + (16 * (i - r_prev) + c16) * LDP + g4)), nyt = neg4(ld4(L.Ys + (16 * i + c16) * LDP + g4));
                        float* ap = At + (size_t)lslot(i, kn) * 256 + lane * 4;
                        st4(ap, xty(nyt, vtj, xty(nvt, ytj, ld4(ap))));
                    }
                    __syncthreads();
                }
                BCLK(7);
                float p0[16], p1[16];
                const int lr0 = tid, lr1 = tid + 256, m = NP - 16 * r_next;
                {
                    const int ra = 16 * r_next + lr0, rb = 16 * r_next + lr1;
                    const float* ta = At + (size_t)lslot(ra >> 4, kn) * 256 + 64 * ((ra & 15) >> 2) + (ra & 3);
                    const float* tb = At + (size_t)lslot(rb >> 4, kn) * 256 + 64 * ((rb & 15) >> 2) + (rb & 3);
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        p0[c] = (lr0 < m) ? ta[4 * c] : 0.f;
                        p1[c] = (lr1 < m) ? tb[4 * c] : 0.f;
                    }
                }
                BCLK(8);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bool b0 = lr0 > j;
                    const float x0 = b0 ? p0[j] : 0.f, x1 = p1[j];
                    float part[16];
#pragma unroll
                    for (int c = 0; c < j; ++c) part[c] = 0.f;
                    part[j] = x0 * x0 + x1 * x1;
#pragma unroll
                    for (int c = j + 1; c < 16; ++c) part[c] = x0 * p0[c] + x1 * p1[c];
                    const float tot = wave_sum16(part, lane);
                    float* rd = L.red + (j & 1) * 68;
                    float* rj = L.rowj + (j & 1) * 16;
                    if ((lane & 3) == 0) rd[4 * (((lane >> 2) & 1) + ((lane >> 3) & 1) * 2 + ((lane >> 4) & 1) * 4 + ((lane >> 5) & 1) * 8) + wave] = tot;        // [column][wave]
                    if (tid == j) {
#pragma unroll
                        for (int c = j; c < 16; ++c) rj[c] = p0[c];
                    }
                    __syncthreads();
                    const f32x4 ssp = ld4(rd + 4 * j);
                    const float ss = (ssp[0] + ssp[1]) + (ssp[2] + ssp[3]);
                    const float alpha = rj[j];
                    float tj = 0.f;
                    if (ss > 0.f) {                                                       // uniform
                        const float nn = alpha * alpha + ss;
                        float norm = nn * __builtin_amdgcn_rsqf(nn);                          // sqrt to 1 ulp + one Newton step: norm <- 0.5 (norm + nn / norm)
                        norm = 0.5f * (norm + nn * __builtin_amdgcn_rcpf(norm));
                        const float beta = alpha >= 0.f ? -norm : norm;
                        float rb = __builtin_amdgcn_rcpf(beta), rs = __builtin_amdgcn_rcpf(alpha - beta);
                        rb = rb * (2.0f - beta * rb);                                         // one Newton step each: <= 1 ulp
                        rs = rs * (2.0f - (alpha - beta) * rs);
                        tj = (beta - alpha) * rb;
                        const float scale = rs;
                        const float v0 = b0 ? x0 * scale : (lr0 == j ? 1.0f : 0.f), v1 = x1 * scale;
                        const float tv0 = tj * v0, tv1 = tj * v1;
#pragma unroll
                        for (int c = j + 1; c < 16; ++c) {
                            const f32x4 dp = ld4(rd + 4 * c);
                            const float w = rj[c] + scale * ((dp[0] + dp[1]) + (dp[2] + dp[3]));
                            p0[c] -= tv0 * w;
                            p1[c] -= tv1 * w;
                        }
                        p0[j] = b0 ? v0 : (lr0 == j ? beta : p0[j]);
                        p1[j] = v1;
                    } else {
                        p0[j] = b0 ? 0.f : p0[j];
                        p1[j] = 0.f;
                    }
                    if (tid == 0) L.taus[j] = tj;
                }
                BCLK(9);
                // V' (unit lower trapezoidal) -> its tiles in the workspace; R -> tile (kn + 1, kn)
                if (lr0 < m) {
                    float* vt = Vg + (size_t)(vb_next + (lr0 >> 4)) * 256 + 64 * ((lr0 & 15) >> 2) + (lr0 & 3);
#pragma unroll
                    for (int c = 0; c < 16; ++c) vt[4 * c] = (lr0 > c) ? p0[c] : (lr0 == c ? 1.0f : 0.f);
                }
                if (lr1 < m) {
                    float* vt = Vg + (size_t)(vb_next + (lr1 >> 4)) * 256 + 64 * ((lr1 & 15) >> 2) + (lr1 & 3);
#pragma unroll
                    for (int c = 0; c < 16; ++c) vt[4 * c] = p1[c];
                }
                if (tid < 16) {
                    float* tl = At + (size_t)lslot(kn + 1, kn) * 256;     // tile (kn + 1, kn)[b][a] = R[b][a]
                    const int bq = tid;
#pragma unroll
                    for (int a = 0; a < 16; ++a) tl[64 * (bq >> 2) + 4 * a + (bq & 3)] = (a >= bq) ? p0[a] : 0.f;
                }
                __syncthreads();
            }
        }
        BCLK(0);
        // ---- the pass over the stored tiles ----
        f32x4 part[BAND_MAXNT];
#pragma unroll
        for (int j = 0; j < BAND_MAXNT; ++j) part[j] = ZERO4;
        f32x4 gp = ZERO4, wu[2] = {ZERO4, ZERO4};
        const int jlo = BACK ? 0 : (have_next ? r_next : r_prev);
        const int ilo = BACK ? (have_next ? r_next : r_prev) : jlo;
        if (want_m || have_next) {
            for (int i = NT - 1; i >= ilo; --i) {
                if (row_owner(NT, i) != wave) continue;
                const bool upd = want_m && have_prev && i >= r_prev, xf = have_next && i >= r_next;
                // (back pass: V and Y sit in LDS as split pairs -- their negation flips the sign bits of both halves)
                const f32x4 nvt = upd ? negop<BACK>(ld4(L.Vs + (16 * (i - r_prev) + c16) * LDP + g4)) : ZERO4;
                const f32x4 nyt = upd ? negop<BACK>(ld4(L.Ys + (16 * i + c16) * LDP + g4)) : ZERO4;
                const f32x4 vni = xf ? bload4(Vr, lane16, (vb_next + i - r_next) * 1024) : ZERO4;
                [[maybe_unused]] const f32x4 vnis = (BACK && xf) ? bload4(Vsr, lane16, (i - r_next) * 1024) : ZERO4;
                if (xf) {
                    if constexpr (!BACK) gp = xty(vni, vni, gp);
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        if (p < CP) wu[p] = xty(vni, ld4(Ut + ((size_t)i * CP + p) * 256 + lane * 4), wu[p]);
                }
                if (!want_m) continue;
                f32x4 xo = ZERO4;
                const int rowbase = lslot(i, 0);
                [[maybe_unused]] const float kap = sV * sY, ikap = isV * isY;
                // Tiles and the next panel's V tiles arrive through rings of PF / 2 registers, refilled PF / 2 tiles ahead.  Every memory instruction of the row
                // sits OUTSIDE the uniform branches (an absent tile reads / writes an offset the descriptor rejects: no traffic), so the in-order memory counter is
                // known statically and a tile's wait is vmcnt(2 PF + ...) instead of vmcnt(0); only arithmetic and LDS traffic are conditional.
                constexpr int PF = BACK ? 4 : 1, VPF = BACK ? 2 : 1;      // (forward: more tiles in flight cost registers -- spills in the QR -- and bought nothing: measured)
                f32x4 ring[PF], vring[VPF];
#pragma unroll
                for (int u = 0; u < PF; ++u) ring[u] = bload4_pol<BACK ? DKT_BAND_BACK_LD_POL : 0>(Ar, lane16, (u >= jlo && u <= i) ? (rowbase + u) * 1024 : OOB);
#pragma unroll
                for (int u = 0; u < VPF; ++u) {
                    const bool vok = have_next && u >= r_next && u < i;
                    vring[u] = BACK ? bload4(Vsr, lane16, vok ? (u - r_next) * 1024 : OOB) : bload4(Vr, lane16, vok ? (vb_next + u - r_next) * 1024 : OOB);
                }
#pragma unroll
                for (int j = 0; j < BAND_MAXNT; ++j) {
                    const bool in = j >= jlo && j <= i;                                     // uniform
                    f32x4 a = ring[j % PF];
                    const f32x4 vnj = vring[j % VPF];
                    ring[j % PF] = bload4_pol<BACK ? DKT_BAND_BACK_LD_POL : 0>(Ar, lane16, (j + PF >= jlo && j + PF <= i) ? (rowbase + j + PF) * 1024 : OOB);
                    {
                        const bool vok = have_next && j + VPF >= r_next && j + VPF < i;
                        vring[j % VPF] = BACK ? bload4(Vsr, lane16, vok ? (j + VPF - r_next) * 1024 : OOB) : bload4(Vr, lane16, vok ? (vb_next + j + VPF - r_next) * 1024 : OOB);
                    }
                    if (in) {
                        if (upd) {
                            if constexpr (BACK) {
                                a *= kap;                                                   // the products below come in units of sV sY
                                a = xtyh(nvt, ld4(L.Ys + (16 * j + c16) * LDP + g4), a);
                                if (j >= r_prev) a = xtyh(nyt, ld4(L.Vs + (16 * (j - r_prev) + c16) * LDP + g4), a);
                                a *= ikap;
                            } else {
                                // (two independent accumulator chains of four MFMAs instead of one of eight)
                                a = xty(nvt, ld4(L.Ys + (16 * j + c16) * LDP + g4), a);
                                if (j >= r_prev) a += xty0(nyt, ld4(L.Vs + (16 * (j - r_prev) + c16) * LDP + g4));
                            }
                        }
                        if (xf) {                                                           // V'_i^T A_ij -> Xt'_j
                            if constexpr (BACK) part[j] = xtyh(vnis, split_h2(a, sA), part[j]);
                            else part[j] = xty(vni, a, part[j]);
                        }
                        if (have_next && j >= r_next && j < i) {
                            // Xt'_i += V'_j^T A_ij^T: the tile transposed through the wave's scratch
                            // (no fence: the LDS serves a wave's instructions in order, and a release fence would also drain the tiles in flight -- vmcnt(0) on
                            //  every tile, measured as one memory latency per tile; the two index patterns may alias, so the compiler keeps their order)
                            __builtin_amdgcn_wave_barrier();
#pragma unroll
                            for (int q = 0; q < 4; ++q) L.Tsc[c16 * TSC_LD + g4 + q] = a[q];
                            __builtin_amdgcn_wave_barrier();
                            f32x4 at;
#pragma unroll
                            for (int q = 0; q < 4; ++q) at[q] = L.Tsc[(g4 + q) * TSC_LD + c16];
                            if constexpr (BACK) xo = xtyh(vnj, split_h2(at, sA), xo);
                            else xo = xty(vnj, at, xo);
                        }
                        if (j == i) part[j] += xo;
                    }
                    bstore4_pol<BACK ? DKT_BAND_BACK_ST_POL : 0>(Ar, a, lane16, (in && upd) ? (rowbase + j) * 1024 : OOB);
                }
            }
        }
        __syncthreads();
        BCLK(1);
        if (have_next) {
            if constexpr (BACK) {
                if (want_m) {
                    const float un = isV * isA;                                             // the accumulators are in units of sV sA
#pragma unroll
                    for (int j = 0; j < BAND_MAXNT; ++j) part[j] *= un;
                }
            }
            // ---- the waves' partials meet in LDS in a fixed order: X' (over the dead Y), V'^T V', V'^T U.  Four rounds; in round r wave w brings its partials of the
            //      tile columns j = (w + r) mod 4 (+ 4, 8, ...): all four waves work in every round, on disjoint columns, and a column sees the waves in the fixed
            //      order j mod 4, j mod 4 - 1, ...  (One wave per round with all its columns -- the other three waiting at the barrier -- was 7 % of the kernel.) ----
            for (int r = 0; r < 4; ++r) {
                if (want_m) {
                    // straight-line code: a tile outside the range goes to the wave's scratch instead (uniform branches around out-of-line bodies cost more than
                    // the LDS round trips: measured 4 k cycles per round)
                    float* const sink = L.Tsc + lane * 4;
                    auto meet = [&](auto sel_c) {
                        constexpr int SEL = decltype(sel_c)::value, CNT = (BAND_MAXNT - SEL + 3) / 4;
                        f32x4 old[CNT];
                        float* xp[CNT];
#pragma unroll
                        for (int u = 0; u < CNT; ++u) {
                            const int j = 4 * u + SEL;
                            xp[u] = (j >= jlo && j < NT) ? L.Ys + (16 * j + c16) * LDP + g4 : sink;
                            old[u] = ld4(xp[u]);
                        }
#pragma unroll
                        for (int u = 0; u < CNT; ++u) st4(xp[u], r == 0 ? part[4 * u + SEL] : part[4 * u + SEL] + old[u]);
                    };
                    switch ((wave + r) & 3) {
                        case 0: meet(std::integral_constant<int, 0>{}); break;
                        case 1: meet(std::integral_constant<int, 1>{}); break;
                        case 2: meet(std::integral_constant<int, 2>{}); break;
                        default: meet(std::integral_constant<int, 3>{}); break;
                    }
                }
                if (wave == r) {
                    if constexpr (!BACK) st4(L.Gs + lane * 4, r == 0 ? gp : gp + ld4(L.Gs + lane * 4));
                    st4(L.Wus + lane * 4, r == 0 ? wu[0] : wu[0] + ld4(L.Wus + lane * 4));
                    st4(L.Wus + 256 + lane * 4, r == 0 ? wu[1] : wu[1] + ld4(L.Wus + 256 + lane * 4));
                }
                __syncthreads();
            }
            BCLK(2);
            // ---- V' -> LDS, T' ----
            for (int i2 = wave; i2 < mtn; i2 += 4) {
                const f32x4 v = ld4(Vg + (size_t)(vb_next + i2) * 256 + lane * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) L.Vs[(16 * i2 + g4 + q) * LDP + c16] = v[q];
            }
            if constexpr (!BACK) {
                // T by rows (larft): T[i][i] = tau_i, T[i][j] = -tau_j sum_{l=i}^{j-1} T[i][l] G[l][j]; thread i < 16 owns row i
                if (tid < 16) {
                    float tr[16];
                    const float ti = L.taus[tid];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        float s = 0.f;
#pragma unroll
                        for (int l = 0; l < j; ++l) s += tr[l] * L.Gs[64 * (l >> 2) + 4 * j + (l & 3)];
                        tr[j] = (j > tid) ? -L.taus[j] * s : (j == tid ? ti : 0.f);
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) L.Ts[tid * LDP + j] = tr[j];
                }
            } else {
                if (wave == 0) {
                    const f32x4 tt = ld4(Tg + (size_t)kn * 256 + lane * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) L.Ts[(g4 + q) * LDP + c16] = tt[q];
                }
            }
            __syncthreads();
            if constexpr (!BACK) {
                if (wave == 0) {
                    f32x4 tt;
#pragma unroll
                    for (int q = 0; q < 4; ++q) tt[q] = L.Ts[(g4 + q) * LDP + c16];
                    st4(Tg + (size_t)kn * 256 + lane * 4, tt);
                }
            }
            BCLK(3);
            // ThT = Th^T in the accumulator layout: forward Th = T^T, back Th = T
            f32x4 ThT;
#pragma unroll
            for (int q = 0; q < 4; ++q) ThT[q] = BACK ? L.Ts[c16 * LDP + g4 + q] : L.Ts[(g4 + q) * LDP + c16];
            // ---- U <- (I - V' Th V'^T) U;  per-wave partial of S = V'^T X' ----
            {
                f32x4 z[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) z[p] = (p < CP) ? xty0(ThT, ld4(L.Wus + p * 256 + lane * 4)) : ZERO4;
                f32x4 sp = ZERO4;
                for (int i2 = wave; i2 < mtn; i2 += 4) {
                    const int i = r_next + i2;
                    const f32x4 nvt = neg4(ld4(L.Vs + (16 * i2 + c16) * LDP + g4));
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        if (p < CP) {
                            float* up = Ut + ((size_t)i * CP + p) * 256 + lane * 4;
                            st4(up, xty(nvt, z[p], ld4(up)));
                        }
                    }
                    if (want_m) {
                        f32x4 xi, vi;
#pragma unroll
                        for (int q = 0; q < 4; ++q) { xi[q] = L.Ys[(16 * i + g4 + q) * LDP + c16]; vi[q] = L.Vs[(16 * i2 + g4 + q) * LDP + c16]; }
                        sp = xty(vi, xi, sp);
                    }
                }
                st4(L.Part + wave * 256 + lane * 4, sp);
            }
            __syncthreads();
            BCLK(4);
            // ---- Wm = Th S Th^T;  Yt_j = Th Xt_j - 0.5 Wm Vt_j  (in place over X') ----
            if (want_m) {
                const f32x4 s = ld4(L.Part + lane * 4) + ld4(L.Part + 256 + lane * 4) + ld4(L.Part + 512 + lane * 4) + ld4(L.Part + 768 + lane * 4);
                const f32x4 tmp = xty0(s, ThT);                   // S Th^T
                f32x4 hwm = xty0(ThT, tmp);                       // Th S Th^T
                hwm *= -0.5f;
                float ymax = 0.f;
                for (int j = (BACK ? 0 : r_next) + wave; j < NT; j += 4) {
                    float* xp = L.Ys + (16 * j + c16) * LDP + g4;
                    f32x4 y = xty0(ThT, ld4(xp));
                    if (j >= r_next) y = xty(hwm, ld4(L.Vs + (16 * (j - r_next) + c16) * LDP + g4), y);
                    st4(xp, y);
                    if constexpr (BACK) ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(y[0]), fabsf(y[1]))), fmaxf(fabsf(y[2]), fabsf(y[3])));
                }
                if constexpr (BACK) {
                    ymax = wave_reduce_dpp<true>(ymax);
                    if (lane == 0) L.red[wave] = ymax;
                }
            }
            __syncthreads();
            if constexpr (BACK) {
                if (want_m) {
                    // ---- V' and Y' become split pairs in place (Y by its own maximum), V' also as split accumulator-layout tiles in the workspace ----
                    const float ym = fmaxf(fmaxf(L.red[0], L.red[1]), fmaxf(L.red[2], L.red[3]));
                    sY = 1.0f; isY = 1.0f;
                    if (ym > 0.f && ym < 1e30f) sY = scale_for(ym, isY);
                    for (int j = wave; j < NT; j += 4) {
                        float* xp = L.Ys + (16 * j + c16) * LDP + g4;
                        st4(xp, split_h2(ld4(xp), sY));
                    }
                    for (int i2 = wave; i2 < mtn; i2 += 4) {
                        float* vp = L.Vs + (16 * i2 + c16) * LDP + g4;
                        st4(vp, split_h2(ld4(vp), sV));
                    }
                    if (it + 1 < npan) {                                                    // the panel after the next: its tiles split, for the next pass' products
                        const int vb2 = band_voff(NT, kn - 1);
                        for (int i2 = wave; i2 < mtn + 1; i2 += 4)
                            bstore4(Vsr, split_h2(ld4(Vg + (size_t)(vb2 + i2) * 256 + lane * 4), sV), lane16, i2 * 1024);
                    }
                    __syncthreads();
                }
            }
            BCLK(5);
        }
    }
#ifdef DKT_BAND_CLOCKS
    // measurement build (tools/band_phase_clocks.py): thread 0's accumulated s_memtime ticks per phase, reported through alpha[b, 0 / 1, 0..7] (overwriting it)
    __syncthreads();
    if (tid == 0) {
        float* cw = t.ws + (size_t)t.bcnt * G.ep_floats + (size_t)bl * 12;
        if constexpr (!BACK) {
            for (int i = 0; i < 12; ++i) cw[i] = clkl[i];
        } else {
            BCLK(6);
            float* al = t.a.alpha + (size_t)(t.b0 + bl) * G.C * N;
            for (int i = 0; i < 12; ++i) { al[i] = cw[i]; al[N + i] = clkl[i]; }
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// After the back transform: W[b], alpha[b], the quadratic form and the hyper-gradients.  Kernels of their own: these are streams (705 KB of W out, 705 KB of E in
// per 420-row episode) which, as the tail of band_sym_kernel<true>, ran at that kernel's two workgroups per CU and in 64-byte pieces (a 16 x 16 tile touches 16 rows)
// -- 17 % of its time (profiles/r06/band_v11_phase_clocks.log).  Here a workgroup owns ONE 16-row stripe of the episode: the stripe of E comes in and the stripe of W
// goes out as whole rows through a [16][NP + 4] image in LDS; the tiles are read from / written to the image in the accumulator layout.
//   quadratic form against the ORIGINAL matrix: the reduction's backward error (a few eps |E|) moves the small eigenvalues of K_c by a relative eps |E| sv / noise,
//   which r^T K^-1 r feels in full (the log-determinant averages it out: measured 1e-6 against 3e-5 ... 9e-5 on class-correlated features, tools/band_mll_model.py).
//   With the residual rho = r - K alpha of the computed alpha, r^T K^-1 r = (r + rho)^T alpha up to second order: one product E a~ per episode (a~ = sv alpha, in A's
//   tiles) restores the accuracy of a direct factorisation.  The stripes' partial sums meet in band_reduce_kernel in a fixed order.
// ------------------------------------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int fin_lds_floats(const int NT) { return 16 * (16 * NT + 4) + 4 * 2 * 256; }

__global__ __launch_bounds__(256, 4) void band_finish_kernel(BandArgs t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const BandGeo& G = t.g;
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g4 = (lane >> 2) & 12;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = G.NT, NP = 16 * NT, N = G.N, CP = G.CP, SLD = NP + 4;
    const int i = blockIdx.x, bl = blockIdx.y, b = t.b0 + bl;
    float* ep = t.ws + (size_t)bl * G.ep_floats;
    const float* Ut = ep + G.oAm;
    const brsrc Ar = mk_rsrc(ep + G.oA, (unsigned)((size_t)NT * (NT + 1) / 2 * 1024));
    const int lane16 = lane * 16;
    float* S = smem;                       // [16][SLD]
    float* Red = smem + 16 * SLD;          // [4 waves][2][256]
    const bool vec = (N & 3) == 0;
    constexpr int MAXT = (BAND_MAXNT + 3) / 4, MAXQ = (16 * 4 * BAND_MAXNT + 255) / 256;       // tiles per wave; 16-byte pieces of the stripe per thread
    // Every memory request of a phase is issued before the first use (a loop of load -> store pairs is one memory round trip per trip: measured, 0.64 ms per 1024
    // episodes of 420 rows for 1.8 GB); absent pieces read an offset the descriptor rejects.
    // ---- rows 16 i .. 16 i + 15 of E -> the image (zero beyond N); the W stripe's tiles (i, j <= i) / (j > i, i) requested alongside ----
    f32x4 wv[MAXT];
    {
        const float* Eb = t.a.E + (size_t)b * N * N;
        if (vec) {
            const brsrc Er = mk_rsrc(Eb, (unsigned)((size_t)N * N * 4));
            const int q4 = NP >> 2;
            f32x4 ev[MAXQ];
#pragma unroll
            for (int u = 0; u < MAXQ; ++u) {
                const int idx = tid + 256 * u, rr = idx / q4, col = 4 * (idx - rr * q4), r = 16 * i + rr;
                ev[u] = bload4(Er, (rr < 16 && r < N && col < N) ? (r * N + col) * 4 : OOB, 0);
            }
#pragma unroll
            for (int u = 0; u < MAXT; ++u) {
                const int j = wave + 4 * u;
                wv[u] = bload4(Ar, lane16, (t.grad && j < NT) ? (j <= i ? lslot(i, j) : lslot(j, i)) * 1024 : OOB);
            }
#pragma unroll
            for (int u = 0; u < MAXQ; ++u) {
                const int idx = tid + 256 * u, rr = idx / q4, col = 4 * (idx - rr * q4);
                if (rr < 16) st4(S + rr * SLD + col, ev[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < MAXT; ++u) {
                const int j = wave + 4 * u;
                wv[u] = bload4(Ar, lane16, (t.grad && j < NT) ? (j <= i ? lslot(i, j) : lslot(j, i)) * 1024 : OOB);
            }
            for (int idx = tid; idx < 16 * NP; idx += 256) {
                const int rr = idx / NP, col = idx - rr * NP, r = 16 * i + rr;
                S[rr * SLD + col] = (r < N && col < N) ? Eb[(size_t)r * N + col] : 0.f;
            }
        }
    }
    // ---- (E a~) for the stripe: tile (j, i) of E in the accumulator layout, element [4g + q][c] = E[16j + 4g + q][16i + c] = E[16i + c][16j + 4g + q] ----
    {
        const brsrc Ur = mk_rsrc(Ut, (unsigned)(NT * CP * 1024));
        f32x4 ut[MAXT][2];
#pragma unroll
        for (int u = 0; u < MAXT; ++u)
#pragma unroll
            for (int p = 0; p < 2; ++p) ut[u][p] = bload4(Ur, lane16, (wave + 4 * u < NT && p < CP) ? ((wave + 4 * u) * CP + p) * 1024 : OOB);
        __syncthreads();
        f32x4 ea[2] = {ZERO4, ZERO4};
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            const int j = wave + 4 * u;
            if (j < NT) {
                const f32x4 e = ld4(S + c16 * SLD + 16 * j + g4);
                ea[0] = xty(e, ut[u][0], ea[0]);
                if (CP > 1) ea[1] = xty(e, ut[u][1], ea[1]);
            }
        }
        st4(Red + (wave * 2 + 0) * 256 + lane * 4, ea[0]);
        st4(Red + (wave * 2 + 1) * 256 + lane * 4, ea[1]);
    }
    __syncthreads();
    if (wave < CP) {                                           // wave p: class column p of the stripe
        const int p = wave, cls = 16 * p + c16;
        const f32x4 eap = (ld4(Red + (0 + p) * 256 + lane * 4) + ld4(Red + (2 + p) * 256 + lane * 4)) + (ld4(Red + (4 + p) * 256 + lane * 4) + ld4(Red + (6 + p) * 256 + lane * 4));
        float qp[4] = {0.f, 0.f, 0.f, 0.f};                   // per class: (2r - E a~ - mu a~).a~,  a~.E a~,  a~.a~,  sum a~
        if (cls < G.C) {
            const f32x4 a = ld4(Ut + ((size_t)i * CP + p) * 256 + lane * 4);
            const float sv = t.a.sv[cls], rs = 1.0f / sv, mu = t.a.noise[cls] * rs, mc = t.a.mean[cls];
            const float* yc = t.a.Y + (size_t)b * t.a.y_bstride + (size_t)cls * N;
            float* al = t.a.alpha + ((size_t)b * G.C + cls) * N;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * i + g4 + q;
                if (r < N) {
                    al[r] = a[q] * rs;                         // alpha[b, c, :] = a_c / sv_c
                    qp[0] += (2.0f * (yc[r] - mc) - eap[q] - mu * a[q]) * a[q];
                    qp[1] += eap[q] * a[q];
                    qp[2] += a[q] * a[q];
                    qp[3] += a[q];
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            qp[v] += __shfl_xor(qp[v], 16, DKT_WAVE);
            qp[v] += __shfl_xor(qp[v], 32, DKT_WAVE);
        }
        if (lane < 16) {
            float* part = ep + G.oG + (size_t)i * 128;        // (the G tiles are dead by now) [stripe][4][32]
#pragma unroll
            for (int v = 0; v < 4; ++v) part[v * 32 + 16 * p + lane] = qp[v];
        }
    }
    if (!t.grad) return;                                       // uniform
    __syncthreads();
    // ---- the stripe of W: tiles (i, j <= i) as stored, tiles (j > i, i) transposed, the diagonal tile symmetrised; then whole rows out ----
    {
        f32x4 hiden;
#pragma unroll
        for (int q = 0; q < 4; ++q) hiden[q] = (g4 + q == c16) ? 0.5f : 0.f;
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            const int j = wave + 4 * u;
            if (j < NT) {
                f32x4 w = wv[u];
                if (j == i) w = xty(w, hiden, 0.5f * w);       // a diagonal tile carries both halves: 0.5 (M_ii + M_ii^T), bitwise symmetric
                if (j <= i) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) S[(g4 + q) * SLD + 16 * j + c16] = w[q];
                } else {
                    st4(S + c16 * SLD + 16 * j + g4, w);      // tile (j, i)[4g + q][c] = W[16i + c][16j + 4g + q]
                }
            }
        }
    }
    __syncthreads();
    {
        float* Wb = t.a.W + (size_t)b * N * N;
        if (vec) {
            const int q4 = N >> 2;
            for (int idx = tid; idx < 16 * q4; idx += 256) {
                const int rr = idx / q4, col = 4 * (idx - rr * q4), r = 16 * i + rr;
                if (r < N) st4(Wb + (size_t)r * N + col, ld4(S + rr * SLD + col));
            }
        } else {
            for (int idx = tid; idx < 16 * N; idx += 256) {
                const int rr = idx / N, col = idx - rr * N, r = 16 * i + rr;
                if (r < N) Wb[(size_t)r * N + col] = S[rr * SLD + col];
            }
        }
    }
}

// the stripes' partial sums -> logp, dsv, dnoise, dmean; one wave per episode
__global__ __launch_bounds__(64) void band_reduce_kernel(BandArgs t) {
    const BandGeo& G = t.g;
    const int tid = threadIdx.x, bl = blockIdx.x, b = t.b0 + bl, N = G.N;
    if (tid >= G.C) return;
    const float* part = t.ws + (size_t)bl * G.ep_floats + G.oG;
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < G.NT; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) sum[v] += part[(size_t)i * 128 + v * 32 + tid];
    const size_t bc = (size_t)b * G.C + tid;
    const float sv = t.a.sv[tid], rs = 1.0f / sv, mu = t.a.noise[tid] * rs;
    const float quad = sum[0] * rs;
    if (t.grad) {
        const float trz = t.a.dnoise[bc];                          // tr (B + mu)^-1 from the class kernel (NaN for a failed class)
        t.a.dsv[bc] = 0.5f * (sum[1] * rs * rs - ((float)N - mu * trz) * rs);      // tr(M E), M = 0.5 (alpha alpha^T - K^-1):  alpha^T E alpha - tr(K^-1 E)
        t.a.dnoise[bc] = 0.5f * (sum[2] * rs * rs - trz * rs);
        t.a.dmean[bc] = sum[3] * rs;
    }
    t.a.logp[bc] -= 0.5f * quad;                                   // the class kernel left -0.5 log det - N/2 log 2 pi there (NaN for a failed class)
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// Per class: the block LDL^T chain of B + mu I, one wave per class matrix
// ------------------------------------------------------------------------------------------------------------------------------------------------
#ifndef DKT_BAND_CLASS_WAVES
#define DKT_BAND_CLASS_WAVES 4          // (5 and 6 waves per SIMD measured equal: 9.69 / 9.69 / 9.68-9.78 ms per 1024 episodes of 420 rows -- the chain is not waiting for occupancy)
#endif
__global__ __launch_bounds__(256, DKT_BAND_CLASS_WAVES) void band_class_kernel(BandArgs t) {
    const BandGeo& G = t.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = G.NT, N = G.N, C = G.C, CP = G.CP;
    const int bl = blockIdx.x, cls = blockIdx.y * 4 + wave;
    if (cls >= C) return;                                   // (no barrier in this kernel)
    Lane ln;
    ln.lane = lane; ln.g = lane >> 4; ln.c = lane & 15;
    ln.g0 = ln.g == 0; ln.g1 = ln.g == 1; ln.g2 = ln.g == 2;
    const int c16 = ln.c, g4 = 4 * ln.g;
    float* ep = t.ws + (size_t)bl * G.ep_floats;
    const float* At = ep + G.oA;
    const float* Ut = ep + G.oU;
    float* Gg = ep + G.oG + (size_t)cls * NT * 256;
    float* Zg = ep + G.oZd + (size_t)cls * NT * 256;
    float* Pg = ep + G.oPi + (size_t)cls * NT * 256;
    float* Zv = ep + G.oZv + (size_t)cls * NT * 16;
    const int b = t.b0 + bl;
    const float sv = t.a.sv[cls], nz = t.a.noise[cls];
    const float mu = nz / sv;
    const int pu = cls >> 4, cu = cls & 15;                                 // U column of this class
    const bool col0 = c16 == 0;

    f32x4 ngt_prev = ZERO4, yprev = ZERO4, st_prev = ZERO4;
    float lsum = 0.f, trb = 0.f;
    int fail_at = 0;
    // (the tiles of step j + 1 are requested at the top of step j: the chain of a class is ~ 27 x 2 dependent steps, a memory round trip per step would double it)
    auto load_st = [&](const int j) {       // S_j^T in the accumulator layout from the stored tile (j + 1, j) = S_j: lane (g, c) register q <- S_j[c][4g + q]
        const float* sp = At + (size_t)lslot(j + 1, j) * 256 + 64 * (c16 >> 2) + 4 * g4 + (c16 & 3);
        return (f32x4){sp[0], sp[4], sp[8], sp[12]};
    };
    f32x4 P_nx = ld4(At + lane * 4), St_nx = NT > 1 ? load_st(0) : ZERO4;
    f32x4 u_nx = col0 ? ld4(Ut + (size_t)pu * 256 + (4 * g4 + cu) * 4) : ZERO4;
    for (int j = 0; j < NT; ++j) {
        f32x4 P = P_nx;
        const f32x4 St_cur = St_nx, u_cur = u_nx;
        {
            const int jn = j + 1 < NT ? j + 1 : j;
            P_nx = ld4(At + (size_t)lslot(jn, jn) * 256 + lane * 4);
            St_nx = jn + 1 < NT ? load_st(jn) : ZERO4;
            u_nx = col0 ? ld4(Ut + ((size_t)jn * CP + pu) * 256 + (4 * g4 + cu) * 4) : ZERO4;
        }
        float dmax = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (g4 + q == c16) {
                trb += (16 * j + c16 < N) ? P[q] : 0.f;                                   // trace B = trace E
                P[q] = (16 * j + c16 < N) ? P[q] + mu : 1.0f;
            }
        }
        if (j > 0) P = xty(ngt_prev, st_prev, P);                                        // - G_{j-1} S_{j-1}^T
#pragma unroll
        for (int q = 0; q < 4; ++q) dmax = (g4 + q == c16) ? fmaxf(dmax, P[q]) : dmax;
        dmax = wave_reduce_dpp<true>(dmax);
        // kappa = 4^k >= the largest diagonal entry (>= every pivot): the sweep wants pivots <= 1
        float kap = 1.0f, ikap = 1.0f;
        if (dmax > 0.f && dmax < 1e30f) {
            const int e = (int)((__float_as_uint(dmax) >> 23) & 0xffu) - 126;        // dmax = f 2^e, 0.5 <= f < 1
            const int k2 = (e + 1) >> 1;                                              // 4^k2 >= 2^e > dmax
            kap = __uint_as_float((unsigned)(127 + 2 * k2) << 23);
            ikap = __uint_as_float((unsigned)(127 - 2 * k2) << 23);
        }
        f32x4 S;
#pragma unroll
        for (int q = 0; q < 4; ++q) S[q] = -P[q] * ikap;
        float x[16], dv;
        sweep_begin(S, x, dv);
        sweep_plain<0, false>(x, dv, ln, 0);
        const f32x4 M = sweep_end(x, ln);
        const bool valid = 16 * j + c16 < N;
        const unsigned long long badm = __ballot(valid && !(dv > 0.f)) & 0xffffull;
        const int first = (int)__builtin_ctzll(badm | 0x10000ull);
        fail_at = (fail_at == 0 && badm != 0) ? 16 * j + first + 1 : fail_at;
        lsum += (valid && ln.g0) ? __builtin_amdgcn_logf(dv * kap) : 0.f;                 // log2
        f32x4 Pinv = xty0(M, M);
        Pinv *= ikap;
        // vectors: column 0 of a tile carries the 16 entries of this class' vector
        f32x4 y = u_cur;
        if (j > 0) y = xty(ngt_prev, yprev, y);                                          // y_j = u_j - G_{j-1} y_{j-1}
        const f32x4 z = xty0(Pinv, y);
        st4(Pg + (size_t)j * 256 + lane * 4, Pinv);
        if (col0) st4(Zv + j * 16 + g4, z);
        if (j + 1 < NT) {
            const f32x4 St = St_cur;
            st_prev = St;
            const f32x4 Gj = xty0(St, Pinv);                                             // S_j P^-1
            ngt_prev = neg4(xty0(Pinv, St));                                             // -(P^-1 S_j^T) = -G_j^T
            st4(Gg + (size_t)j * 256 + lane * 4, Gj);
        }
        yprev = y;
    }
    // ---- backward: a_j = z_j - G_j^T a_{j+1},  Z_jj = P_j^-1 + G_j^T Z_{j+1,j+1} G_j ----
    float* Am = ep + G.oAm;
    float* AmT = ep + G.oAmT;
    f32x4 a_next = ZERO4, zd_next = ZERO4;
    float trz = 0.f, aa = 0.f;
    f32x4 Pi_nx = ld4(Pg + (size_t)(NT - 1) * 256 + lane * 4), z_nx = col0 ? ld4(Zv + (NT - 1) * 16 + g4) : ZERO4, G_nx = ZERO4;
    for (int j = NT - 1; j >= 0; --j) {
        const f32x4 Pinv = Pi_nx, z = z_nx, Gj = G_nx;
        {
            const int jn = j > 0 ? j - 1 : 0;
            Pi_nx = ld4(Pg + (size_t)jn * 256 + lane * 4);
            z_nx = col0 ? ld4(Zv + jn * 16 + g4) : ZERO4;
            G_nx = ld4(Gg + (size_t)jn * 256 + lane * 4);              // (G_{NT-1} does not exist and is not read: the loop starts with the zero tile)
        }
        f32x4 a, zd;
        if (j == NT - 1) {
            a = z;
            zd = Pinv;
        } else {
            a = xty(neg4(Gj), a_next, z);
            zd = xty(Gj, xty0(zd_next, Gj), Pinv);
        }
        st4(Zg + (size_t)j * 256 + lane * 4, zd);
        if (col0) {
            st4(Am + ((size_t)j * CP + pu) * 256 + (4 * g4 + cu) * 4, a);
            float* at = AmT + ((size_t)pu * NT + j) * 256 + 64 * (cu >> 2) + (cu & 3);
#pragma unroll
            for (int q = 0; q < 4; ++q) { at[4 * (g4 + q)] = a[q]; aa += a[q] * a[q]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) trz += (g4 + q == c16 && 16 * j + c16 < N) ? zd[q] : 0.f;
        a_next = a;
        zd_next = zd;
    }
    lsum = wave_reduce_dpp<false>(lsum);
    trz = wave_reduce_dpp<false>(trz);
    aa = wave_reduce_dpp<false>(aa);
    trb = wave_reduce_dpp<false>(trb);
    // The reduction's backward error (a few eps |E|) is amplified by sv |E| / noise in alpha and W (the quadratic form is repaired from the residual, the log-determinant
    // averages it out).  A class whose a-priori condition bound 1 + sv trace(E) / noise exceeds BAND_KAPPA_MAX hands its episode to the generic kernel's fix-up launch
    // (exact fp32 on the matrix itself, jitter ladder included) -- the reference's frozen noise 0.1 and unit-norm rows stay below it for outputscales up to ~ 4.5 at N = 420.
    if (fail_at == 0 && !(1.0f + sv * trb / nz <= BAND_KAPPA_MAX)) fail_at = -1;
    if (lane == 0) {
        const size_t bc = (size_t)b * C + cls;
        const float qnan = __int_as_float(0x7fc00000);
        const bool ok = fail_at == 0;
        const float logdet = (float)N * logf(sv) + lsum * 0.69314718055994530942f;
        t.a.logp[bc] = ok ? -0.5f * logdet - (float)N * DKT_HALF_LOG_2PI : qnan;       // the quadratic form is added by the back kernel, from the residual of alpha
        t.a.jitter_used[bc] = 0.f;
        t.a.info[bc] = fail_at;
        // the hyper-gradients are finished by the back kernel from alpha in the original coordinates (alpha^T E alpha, alpha^T alpha against the ORIGINAL E);
        // tr (B + mu)^-1 travels there in dnoise[]
        if (t.grad) t.a.dnoise[bc] = ok ? trz : qnan;
        ep[G.oSc + cls] = aa;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// M = sum_c 0.5 cw_c (a_c a_c^T / sv_c - Z^c), Z^c = (B + mu_c)^-1: block column i of Z^c follows from its diagonal block upwards, Z_ji = -G_j^T Z_{j+1,i}.
// Workgroup = episode, 8 waves; a wave owns FOUR block columns -- by distance e from the last one: e = w and 31 - w (accumulator array P, filled from both ends) and
// e = 15 - w and 16 + w (array Q): 46 .. 50 tiles per wave -- and keeps their tiles as accumulators over the classes; the G tiles of TWO classes per step are staged
// through LDS (global_load_lds, double buffered, one barrier per pair), the diagonal blocks come straight from memory.
// ------------------------------------------------------------------------------------------------------------------------------------------------
constexpr int CHAIN_WAVES = 8;
constexpr int NACC_P = BAND_MAXNT, NACC_Q = 2 * BAND_MAXNT - 31 + 1;

// one step of a column chain: z <- G_j^T z (the G tile from the LDS image), accumulate with the alternating sign
__device__ __forceinline__ void chain_step(f32x4& z, f32x4& a, const float* gtile, const float sg) {
    z = xty0(ld4(gtile), z);
    a += sg * z;
}

__global__ __launch_bounds__(64 * CHAIN_WAVES) void band_chain_kernel(BandArgs t) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const BandGeo& G = t.g;
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g4 = (lane >> 2) & 12;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = G.NT, C = G.C, CP = G.CP;
    const int bl = blockIdx.x;
    float* ep = t.ws + (size_t)bl * G.ep_floats;
    float* At = ep + G.oA;
    const float* Gg = ep + G.oG;
    const float* Zg = ep + G.oZd;
    const float* AmT = ep + G.oAmT;
    const int b = t.b0 + bl;
    const int32_t* info = t.a.info + (size_t)b * C;
    // the wave's four block columns, by distance from the last one: e = w, 31 - w (accumulators P: the first from the front, the second from the back) and
    // e = 15 - w, 16 + w (accumulators Q); column i = NT - 1 - e has i + 1 tiles (j = i .. 0), a negative i means "none".  46 .. 50 tiles per wave.
    const int i0 = NT - 1 - wave, i3 = NT - 32 + wave, i1 = NT - 16 + wave, i2 = NT - 17 - wave;
    const int l0 = i0 + 1, l3 = i3 >= 0 ? i3 + 1 : 0, l1 = i1 >= 0 ? i1 + 1 : 0, l2 = i2 >= 0 ? i2 + 1 : 0;
    f32x4 P[NACC_P], Q[NACC_Q];
#pragma unroll
    for (int s = 0; s < NACC_P; ++s) P[s] = ZERO4;
#pragma unroll
    for (int s = 0; s < NACC_Q; ++s) Q[s] = ZERO4;
    const int ngt = NT - 1;                                        // G tiles per class
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(1))) const unsigned char glb_u8;
    // the G tiles of class c straight into LDS buffer bufi (global_load_lds_dwordx4: no staging registers), tiles wave, wave + 8, ...
    auto fetch = [&](const int c, const int bufi) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tile = wave + CHAIN_WAVES * u;
            if (tile < ngt)
                __builtin_amdgcn_global_load_lds((glb_u8*)(Gg + ((size_t)c * NT + tile) * 256) + lane * 16,
                                                 (lds_u8*)(smem + ((size_t)bufi * BAND_MAXNT + tile) * 256), 16, 0, 0);
        }
    };
    // TWO classes per iteration: their chains of one block column have the same length and feed the same accumulator, so every step is one basic block with two
    // independent dependency chains (one class at a time: ~ 350 cycles per step for 4 dependent MFMAs + an LDS read, measured 31 k cycles per class).  LDS: 2 x 2 class
    // images, double buffered.
    const int npair = (C + 1) / 2;
    auto fetch2 = [&](const int pr, const int half) {
        fetch(2 * pr, 2 * half);
        if (2 * pr + 1 < C) fetch(2 * pr + 1, 2 * half + 1);
    };
    fetch2(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0070 | 0x0f00);                   // vmcnt(0)
    __syncthreads();
    for (int pr = 0; pr < npair; ++pr) {
        const int c0 = 2 * pr, c1 = (2 * pr + 1 < C) ? 2 * pr + 1 : c0;       // (an odd class count: the last iteration runs its class twice, the copy with weight 0)
        const float* g0 = smem + (size_t)(2 * (pr & 1)) * (BAND_MAXNT * 256) + lane * 4;
        const float* g1 = (2 * pr + 1 < C) ? g0 + BAND_MAXNT * 256 : g0;
        if (pr + 1 < npair) fetch2(pr + 1, (pr + 1) & 1);
        // (a failed class leaves the episode to the fix-up launch: weight 0 here, whatever its tiles hold)
        const float w0 = info[c0] == 0 ? -0.5f * (t.a.cls_weight ? t.a.cls_weight[c0] : 1.0f) : 0.f;
        const float w1 = (2 * pr + 1 < C && info[c1] == 0) ? -0.5f * (t.a.cls_weight ? t.a.cls_weight[c1] : 1.0f) : 0.f;
        const float* z0p = Zg + (size_t)c0 * NT * 256 + lane * 4;
        const float* z1p = Zg + (size_t)c1 * NT * 256 + lane * 4;
#define DKT_CHAIN_COL(ACC, IDX0, IDXS, I, LEN, NACC)                                                                                   \
        if ((LEN) > 0) {                                                                                                               \
            f32x4 za = ld4(z0p + (size_t)(I) * 256), zb = ld4(z1p + (size_t)(I) * 256);                                                \
            ACC[IDX0] += w0 * za + w1 * zb;                                                                                            \
            _Pragma("unroll") for (int s = 1; s < (NACC); ++s) {                                                                      \
                if (s < (LEN)) {                                                                                                       \
                    za = xty0(ld4(g0 + (size_t)((I) - s) * 256), za);                                                                  \
                    zb = xty0(ld4(g1 + (size_t)((I) - s) * 256), zb);                                                                  \
                    const f32x4 sum = w0 * za + w1 * zb;                       /* Z_ji = (-1)^s G_j^T ... G_{i-1}^T Z_ii */             \
                    ACC[IDXS] += (s & 1) ? -sum : sum;                                                                                 \
                }                                                                                                                      \
            }                                                                                                                          \
        }
        DKT_CHAIN_COL(P, 0, s, i0, l0, NACC_P)
        DKT_CHAIN_COL(P, NACC_P - 1, NACC_P - 1 - s, i3, l3, NACC_P)
        DKT_CHAIN_COL(Q, 0, s, i1, l1, NACC_Q)
        DKT_CHAIN_COL(Q, NACC_Q - 1, NACC_Q - 1 - s, i2, l2, NACC_Q)
#undef DKT_CHAIN_COL
        __builtin_amdgcn_s_waitcnt(0x0070 | 0x0f00);               // vmcnt(0): the next pair's tiles have landed
        __syncthreads();
    }
    // rank-C term M_ji += sum_p (A^T_pj)^T diag(ka) A^T_pi, then the stored triangle: tile (i, j) = M_ji^T
    f32x4 iden;
#pragma unroll
    for (int q = 0; q < 4; ++q) iden[q] = (g4 + q == c16) ? 1.0f : 0.f;
    f32x4 ka[2];                                                   // 0.5 cw_c / sv_c by the class rows of an A^T tile (0 beyond C and for a failed class)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 16 * p + g4 + q;
            ka[p][q] = (c < C && info[c < C ? c : 0] == 0) ? 0.5f * (t.a.cls_weight ? t.a.cls_weight[c] : 1.0f) / t.a.sv[c] : 0.f;
        }
    auto finish = [&](f32x4 m, const int i, const int s, const f32x4 (&ai)[2]) {       // + the rank-C term, store tile (i, i - s) = M_ji^T
        const int j = i - s;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            if (p < CP) m = xty(ld4(AmT + ((size_t)p * NT + j) * 256 + lane * 4), ai[p], m);
        st4(At + (size_t)lslot(i, j) * 256 + lane * 4, s > 0 ? xty0(m, iden) : m);
    };
    auto col_ops = [&](const int i, f32x4 (&ai)[2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            ai[p] = (p < CP && i >= 0) ? ld4(AmT + ((size_t)p * NT + i) * 256 + lane * 4) : ZERO4;
            ai[p] *= ka[p];
        }
    };
    f32x4 a0[2], a1[2];
    col_ops(i0, a0);
    col_ops(i3, a1);
#pragma unroll
    for (int s = 0; s < NACC_P; ++s) {
        if (s < l0) finish(P[s], i0, s, a0);
        if (s < l3) finish(P[NACC_P - 1 - s], i3, s, a1);
    }
    col_ops(i1, a0);
    col_ops(i2, a1);
#pragma unroll
    for (int s = 0; s < NACC_Q; ++s) {
        if (s < l1) finish(Q[s], i1, s, a0);
        if (s < l2) finish(Q[NACC_Q - 1 - s], i2, s, a1);
    }
}

}  // namespace

bool dkt_mll_band_supports(int N, unsigned flags, int C) {
    if (flags & (DKT_MLL_WANT_CHOL | DKT_MLL_E_PER_CLASS | DKT_MLL_FORCE_GENERIC | DKT_MLL_FORCE_BLOCKED | DKT_MLL_FORCE_F32MFMA | DKT_MLL_FORCE_TILED)) return false;
    return N >= 128 && (N + 15) / 16 <= BAND_MAXNT && C >= 2 && C <= 32;
}

// The default dispatch: from 12 classes and 192 episodes per call (tools/band_crossover.py, profiles/r06/band_crossover.log: the reduction costs about the same for 8 and for
// 32 classes, the tile arrays grow with C; below 192 episodes the per-episode chain of 2 x 25 panels does not fill the GPU and the tile arrays' 20 x more workgroups win).
bool dkt_mll_band_applies(int B, int C, int N, unsigned flags) {
    if (!dkt_mll_band_supports(N, flags, C)) return false;
    if (flags & DKT_MLL_FORCE_BAND) return true;
    return C >= 12 && B >= 192;
}

size_t dkt_mll_band_workspace_bytes(int B, int C, int N) {
    const BandGeo g = band_geo(N, C);
    const int bc = B < BAND_CHUNK ? B : BAND_CHUNK;
    size_t fl = (size_t)bc * g.ep_floats + 12 * (size_t)bc + 64;      // (+ 12 floats per episode: the phase clocks of the measurement build)
    const size_t gen = dkt_mll_generic_global_floats(bc, N);               // the fix-up pass works in the same region
    return (fl > gen ? fl : gen) * sizeof(float);
}

int dkt_mll_band_launch(const MllArgs& a, void* workspace, size_t ws_bytes, hipStream_t st) {
    if (!workspace || ws_bytes < dkt_mll_band_workspace_bytes(a.B, a.C, a.N)) return DKT_ERR_WORKSPACE;
    BandArgs t;
    t.a = a;
    t.g = band_geo(a.N, a.C);
    t.ws = (float*)workspace;
    t.grad = (a.flags & DKT_MLL_WANT_GRAD) ? 1 : 0;
    const int NT = t.g.NT;
    const size_t lds = (size_t)sym_lds_floats(NT) * sizeof(float);
    // (twins library: DKT_PAD_BAND_FWD / _BACK / _CLASS = bytes of untouched dynamic LDS on top: one workgroup per CU instead of two -- the occupancy A/B of MEASUREMENTS R6d)
    auto band_pad = [](const size_t base, const char* name) { const size_t v = base + dkt_lds_pad(name); return v > 98304 ? (size_t)98304 : v; };
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)band_sym_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304) != hipSuccess) return DKT_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)band_sym_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304) != hipSuccess) return DKT_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)band_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * BAND_MAXNT * 1024) != hipSuccess) return DKT_ERR_LAUNCH;
        attr_done = true;
    }
    const int Bc = a.B < BAND_CHUNK ? a.B : BAND_CHUNK;
    for (int b0 = 0; b0 < a.B; b0 += Bc) {
        const int bcnt = (a.B - b0 < Bc) ? a.B - b0 : Bc;
        t.b0 = b0;
        t.bcnt = bcnt;
        const int slots = NT * (NT + 1) / 2 + 3 * NT * t.g.CP;
        hipLaunchKernelGGL(band_init_kernel, dim3((slots + 3) / 4, bcnt), dim3(256), 0, st, t);
        hipLaunchKernelGGL(band_sym_kernel<false>, dim3(bcnt), dim3(256), band_pad(lds, "DKT_PAD_BAND_FWD"), st, t);
        hipLaunchKernelGGL(band_class_kernel, dim3(bcnt, (a.C + 3) / 4), dim3(256), dkt_lds_pad("DKT_PAD_BAND_CLASS"), st, t);
        if (t.grad) hipLaunchKernelGGL(band_chain_kernel, dim3(bcnt), dim3(64 * CHAIN_WAVES), 4 * BAND_MAXNT * 1024, st, t);
        hipLaunchKernelGGL(band_sym_kernel<true>, dim3(bcnt), dim3(256), band_pad(lds, "DKT_PAD_BAND_BACK"), st, t);
#ifndef DKT_BAND_CLOCKS                       // (the measurement build reports its phase clocks through alpha)
        hipLaunchKernelGGL(band_finish_kernel, dim3(NT, bcnt), dim3(256), fin_lds_floats(NT) * sizeof(float), st, t);
        hipLaunchKernelGGL(band_reduce_kernel, dim3(bcnt), dim3(64), 0, st, t);
#endif
        MllArgs f = a;
        f.only_failed = a.info;
        dkt_mll_generic_global_launch(f, b0, bcnt, t.ws, st);
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
