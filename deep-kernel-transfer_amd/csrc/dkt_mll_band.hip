// dkt_mll_band.hip -- exact-GP marginal likelihood of the C one-vs-rest models of an episode that SHARE their base matrix (linear / cossim / bncossim,
// 128 <= N <= 447, 8 <= C <= 32: the 20-way shapes of train.py:132-133) through ONE orthogonal reduction per episode instead of C factorisations.
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
// 4 x v_mfma_f32_16x16x4_f32 (dkt_mfma_tiles.h) -- exact fp32 throughout, no split, no range contract.  Kernels (workspace per episode in `BandGeo`):
//   band_init_kernel      E[b] -> full tile array A (both triangles), U = [y_c - m_c] as tiles
//   band_twosided_kernel<false>  (workgroup = episode)  panels k = 0 .. NT-3: Householder QR of block column k below the band (rows over threads, one
//                         exchange per column), T_k by the larft recurrence from V^T V, U <- H^T U, and the two-sided update A <- H^T A H as
//                         X = A V, S = V^T X, Y = X Th^T - 0.5 V (Th S Th^T), A -= V Y^T + Y V^T with V and Y in LDS
//   band_class_kernel     (wave = class matrix)  block LDL^T chain of B + mu_c I on the diagonal-tile sweep of dkt_mfma_tiles.h: pivots, P_j^-1, G_j = S_j P_j^-1,
//                         forward / backward substitution for a_c, Z_jj = diagonal blocks of (B + mu_c)^-1, all scalars of the class
//   band_chain_kernel     (wave = block columns)  Z_ji = -G_j^T Z_{j+1,i} upwards from the diagonal, accumulated over the classes in registers, + the rank-C term
//   band_twosided_kernel<true>   M <- H_k M H_k^T for k = NT-3 .. 0 (the same code with Th = T), a <- H_k a; then W[b] and alpha[b] are stored
// Attempt 0 only (no jitter): an episode with a failed class is redone -- jitter ladder and all -- by the generic kernel's fix-up launch, as in the tile-array path.
#include "dkt_mfma_tiles.h"

namespace {

using namespace dkt_mfma;

constexpr int BAND_MAXNT = 28;             // N <= 447
constexpr int LDP = 20;                    // row stride (floats) of the panel-shaped LDS arrays: 16-byte aligned rows
constexpr int BAND_CHUNK = 1024;           // episodes per pass over the workspace

struct BandGeo {
    int N, NT, C, CP;                      // CP = class-column tiles of U / A
    int oA, oV, oT, oU, oAm, oAmT, oG, oZd, oPi, oZv;      // offsets (floats) into the episode's workspace
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
    g.oA = take(g.NT * g.NT);
    g.oV = take(g.NT * (g.NT - 1) / 2);
    g.oT = take(g.NT);
    g.oU = take(g.NT * g.CP);
    g.oAm = take(g.NT * g.CP);
    g.oAmT = take(g.NT * g.CP);
    g.oG = take(C * g.NT);
    g.oZd = take(C * g.NT);
    g.oPi = take(C * g.NT);
    g.oZv = take((C * g.NT + 15) / 16);     // 16 floats per (class, block)
    g.ep_floats = o;
    return g;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 neg4(const f32x4 v) { return (f32x4){-v[0], -v[1], -v[2], -v[3]}; }
constexpr f32x4 ZERO4 = {0.f, 0.f, 0.f, 0.f};

// ------------------------------------------------------------------------------------------------------------------------------------------------
// E[b] -> A tiles (full: both triangles), U tiles
// ------------------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_init_kernel(BandArgs t) {
    const BandGeo& G = t.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c16 = lane & 15, g4 = (lane >> 2) & 12;
    const int NT = G.NT, N = G.N, C = G.C;
    const int slot = blockIdx.x * 4 + wave, bl = blockIdx.y;
    float* ep = t.ws + (size_t)bl * G.ep_floats;
    const int b = t.b0 + bl;
    if (slot < NT * NT) {
        const int i = slot / NT, j = slot - i * NT;
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
    } else if (slot < NT * NT + NT * G.CP) {
        const int u = slot - NT * NT, i = u / G.CP, p = u - i * G.CP;
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
    } else if (slot < NT * NT + 3 * NT * G.CP) {
        // A / A^T tiles: the class kernel writes the columns of its classes only; the others (the padding) must read as zero
        st4(ep + G.oAm + (size_t)(slot - NT * NT - NT * G.CP) * 256 + lane * 4, ZERO4);       // (oAmT follows oAm)
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// The two-sided kernel: BACK = false: reduction of A to block-tridiagonal form (panels in order); BACK = true: M <- Q M Q^T (panels in reverse)
// ------------------------------------------------------------------------------------------------------------------------------------------------
struct TsLds {
    float* Vs;       // [NP - 16][LDP]  V of the panel, row-major, panel-local rows
    float* Xs;       // [NP][LDP]       X, then Y, rows = global rows; aliased by the partial tiles of the G / V^T U reductions before X exists
    float* Ts;       // [16][LDP]       T row-major
    float* Part;     // [4][256]        per-wave partial tile (S)
    float* Gs;       // [256]           reduced tile (G, then S is reduced in registers)
    float* Wus;      // [2][256]        V^T U reduced
    float* red;      // [2][4][17]      QR exchange
    float* rowj;     // [2][16]
};

__host__ __device__ inline int ts_lds_floats(const int NT) {
    const int NP = 16 * NT;
    const int xs = NP * LDP > 3072 ? NP * LDP : 3072;            // the X region also holds 4 x 3 partial tiles before X exists
    return (NP - 16) * LDP + xs + 16 * LDP + 4 * 256 + 256 + 2 * 256 + 2 * 4 * 17 + 2 * 16 + 8;
}

#ifdef DKT_BAND_CLOCKS
#define BCLK(i) do { __builtin_amdgcn_sched_barrier(0); if (tid == 0) clk[i] += __builtin_amdgcn_s_memtime() - tlast; tlast = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define BCLK(i) do { } while (0)
#endif

template <bool BACK>
__global__ __launch_bounds__(256, 2) void band_twosided_kernel(BandArgs t) {
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
    TsLds L;
    L.Vs = smem;
    L.Xs = L.Vs + (NP - 16) * LDP;
    L.Ts = L.Xs + (NP * LDP > 3072 ? NP * LDP : 3072);
    L.Part = L.Ts + 16 * LDP;
    L.Gs = L.Part + 4 * 256;
    L.Wus = L.Gs + 256;
    L.red = L.Wus + 2 * 256;
    L.rowj = L.red + 2 * 4 * 17;
#ifdef DKT_BAND_CLOCKS
    unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    const bool want_m = !BACK || t.grad;

    for (int kk = 0; kk < NT - 2; ++kk) {
        const int k = BACK ? NT - 3 - kk : kk;
        const int r0 = 16 * (k + 1), m = NP - r0, mt = NT - k - 1;
        float tau[16];
        if constexpr (!BACK) {
            // ---- Householder QR of the panel: thread t owns the panel rows t and t + 256 (16 columns each in registers) ----
            float p0[16], p1[16];
            const int lr0 = tid, lr1 = tid + 256;
            {
                const int ra = r0 + lr0, rb = r0 + lr1;
                const float* ta = At + ((size_t)(ra >> 4) * NT + k) * 256 + 64 * ((ra & 15) >> 2) + (ra & 3);
                const float* tb = At + ((size_t)(rb >> 4) * NT + k) * 256 + 64 * ((rb & 15) >> 2) + (rb & 3);
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    p0[c] = (lr0 < m) ? ta[4 * c] : 0.f;
                    p1[c] = (lr1 < m) ? tb[4 * c] : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool b0 = lr0 > j;
                const float x0 = b0 ? p0[j] : 0.f, x1 = p1[j];
                float part[16];
                part[j] = x0 * x0 + x1 * x1;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) part[c] = x0 * p0[c] + x1 * p1[c];
#pragma unroll
                for (int c = j; c < 16; ++c) part[c] = wave_reduce_dpp<false>(part[c]);
                float* rd = L.red + (j & 1) * 68;
                float* rj = L.rowj + (j & 1) * 16;
                if (lane == 0) {
#pragma unroll
                    for (int c = j; c < 16; ++c) rd[wave * 17 + c] = part[c];
                }
                if (tid == j) {
#pragma unroll
                    for (int c = j; c < 16; ++c) rj[c] = p0[c];
                }
                __syncthreads();
                const float ss = rd[j] + rd[17 + j] + rd[34 + j] + rd[51 + j];
                const float alpha = rj[j];
                float tj = 0.f;
                if (ss > 0.f) {                                                       // uniform
                    const float norm = sqrtf(alpha * alpha + ss);
                    const float beta = alpha >= 0.f ? -norm : norm;
                    tj = (beta - alpha) / beta;
                    const float scale = 1.0f / (alpha - beta);
                    const float v0 = b0 ? x0 * scale : (lr0 == j ? 1.0f : 0.f), v1 = x1 * scale;
                    const float tv0 = tj * v0, tv1 = tj * v1;
#pragma unroll
                    for (int c = j + 1; c < 16; ++c) {
                        const float w = rj[c] + scale * (rd[c] + rd[17 + c] + rd[34 + c] + rd[51 + c]);
                        p0[c] -= tv0 * w;
                        p1[c] -= tv1 * w;
                    }
                    p0[j] = b0 ? v0 : (lr0 == j ? beta : p0[j]);
                    p1[j] = v1;
                } else {
                    p0[j] = b0 ? 0.f : p0[j];
                    p1[j] = 0.f;
                }
                tau[j] = tj;
            }
            // V -> LDS (unit lower trapezoidal), R -> the band tiles (k, k+1) = R^T and (k+1, k) = R
            if (lr0 < m) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int c = 4 * c4 + q; v[q] = (lr0 > c) ? p0[c] : (lr0 == c ? 1.0f : 0.f); }
                    st4(L.Vs + lr0 * LDP + 4 * c4, v);
                }
            }
            if (lr1 < m) {
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) st4(L.Vs + lr1 * LDP + 4 * c4, (f32x4){p1[4 * c4], p1[4 * c4 + 1], p1[4 * c4 + 2], p1[4 * c4 + 3]});
            }
            if (tid < 16) {
                float* tu = At + ((size_t)k * NT + k + 1) * 256;       // tile (k, k+1)[a][b] = R[b][a]
                float* tl = At + ((size_t)(k + 1) * NT + k) * 256;     // tile (k+1, k)[b][a] = R[b][a]
                const int bq = tid;
#pragma unroll
                for (int a = 0; a < 16; ++a) {
                    const float r = (a >= bq) ? p0[a] : 0.f;
                    tu[64 * (a >> 2) + 4 * bq + (a & 3)] = r;
                    tl[64 * (bq >> 2) + 4 * a + (bq & 3)] = r;
                }
            }
            __syncthreads();
            // V tiles -> global (the back transform reads them)
            for (int it = wave; it < mt; it += 4) {
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = L.Vs[(16 * it + g4 + q) * LDP + c16];
                st4(Vg + (size_t)(band_voff(NT, k) + it) * 256 + lane * 4, v);
            }
        } else {
            // ---- V_k, T_k from the workspace ----
            for (int it = wave; it < mt; it += 4) {
                const f32x4 v = ld4(Vg + (size_t)(band_voff(NT, k) + it) * 256 + lane * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) L.Vs[(16 * it + g4 + q) * LDP + c16] = v[q];
            }
            if (wave == 0) {
                const f32x4 tt = ld4(Tg + (size_t)k * 256 + lane * 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) L.Ts[(g4 + q) * LDP + c16] = tt[q];
            }
            __syncthreads();
        }
        BCLK(0);
        // ---- per-wave partial tiles of G = V^T V (forward: T is built from it) and Wu = V^T U, into the X region (not live yet) ----
        {
            f32x4 gp = ZERO4, wu[2] = {ZERO4, ZERO4};
            for (int it = wave; it < mt; it += 4) {
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = L.Vs[(16 * it + g4 + q) * LDP + c16];
                if constexpr (!BACK) gp = xty(v, v, gp);
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    if (p < CP) wu[p] = xty(v, ld4(Ut + ((size_t)(k + 1 + it) * CP + p) * 256 + lane * 4), wu[p]);
            }
            float* pp = L.Xs + wave * 768;
            st4(pp + lane * 4, gp);
            st4(pp + 256 + lane * 4, wu[0]);
            st4(pp + 512 + lane * 4, wu[1]);
        }
        __syncthreads();
        {
            const float* pp = L.Xs;
            if constexpr (!BACK) L.Gs[tid] = pp[tid] + pp[768 + tid] + pp[1536 + tid] + pp[2304 + tid];
            L.Wus[tid] = pp[256 + tid] + pp[768 + 256 + tid] + pp[1536 + 256 + tid] + pp[2304 + 256 + tid];
            L.Wus[256 + tid] = pp[512 + tid] + pp[768 + 512 + tid] + pp[1536 + 512 + tid] + pp[2304 + 512 + tid];
        }
        __syncthreads();
        if constexpr (!BACK) {
            // T by rows (larft): T[i][i] = tau_i, T[i][j] = -tau_j sum_{l=i}^{j-1} T[i][l] G[l][j]; thread i < 16 owns row i
            if (tid < 16) {
                float tr[16];
                float ti = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) ti = (tid == j) ? tau[j] : ti;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float s = 0.f;
#pragma unroll
                    for (int l = 0; l < j; ++l) s += tr[l] * L.Gs[64 * (l >> 2) + 4 * j + (l & 3)];
                    tr[j] = (j > tid) ? -tau[j] * s : (j == tid ? ti : 0.f);
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) L.Ts[tid * LDP + j] = tr[j];
            }
            __syncthreads();
            if (wave == 0) {
                f32x4 tt;
#pragma unroll
                for (int q = 0; q < 4; ++q) tt[q] = L.Ts[(g4 + q) * LDP + c16];
                st4(Tg + (size_t)k * 256 + lane * 4, tt);
            }
        }
        // ThT = Th^T in the accumulator layout: forward Th = T^T, back Th = T
        f32x4 ThT;
#pragma unroll
        for (int q = 0; q < 4; ++q) ThT[q] = BACK ? L.Ts[c16 * LDP + g4 + q] : L.Ts[(g4 + q) * LDP + c16];
        // ---- U <- (I - V Th V^T) U :  Z_p = Th Wu_p,  U_ip -= V_i Z_p ----
        {
            f32x4 z[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) z[p] = (p < CP) ? xty0(ThT, ld4(L.Wus + p * 256 + lane * 4)) : ZERO4;
            for (int it = wave; it < mt; it += 4) {
                const f32x4 nvt = neg4(ld4(L.Vs + (16 * it + c16) * LDP + g4));
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    if (p < CP) {
                        float* up = Ut + ((size_t)(k + 1 + it) * CP + p) * 256 + lane * 4;
                        st4(up, xty(nvt, z[p], ld4(up)));
                    }
                }
            }
        }
        BCLK(1);
        if (want_m) {
            const int ilo = BACK ? 0 : k + 1;
            // ---- Xt_i = sum_{j > k} V_j^T A_ji  -> Xs (row-major X), per-wave partial of S = V^T X ----
            __syncthreads();                                  // the partial tiles in the X region have been consumed
            f32x4 sp = ZERO4;
            for (int i = ilo + wave; i < NT; i += 4) {
                f32x4 acc0 = ZERO4, acc1 = ZERO4;
                const float* acol = At + (size_t)i * 256 + lane * 4;
                int jt = 0;
                for (; jt + 1 < mt; jt += 2) {
                    f32x4 v0, v1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v0[q] = L.Vs[(16 * jt + g4 + q) * LDP + c16]; v1[q] = L.Vs[(16 * jt + 16 + g4 + q) * LDP + c16]; }
                    const f32x4 a0 = ld4(acol + (size_t)(k + 1 + jt) * NT * 256), a1 = ld4(acol + (size_t)(k + 2 + jt) * NT * 256);
                    acc0 = xty(v0, a0, acc0);
                    acc1 = xty(v1, a1, acc1);
                }
                if (jt < mt) {
                    f32x4 v0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v0[q] = L.Vs[(16 * jt + g4 + q) * LDP + c16];
                    acc0 = xty(v0, ld4(acol + (size_t)(k + 1 + jt) * NT * 256), acc0);
                }
                acc0 += acc1;
                st4(L.Xs + (16 * i + c16) * LDP + g4, acc0);                 // Xt_i[a][b] = X[16 i + b][a]
                if (i > k) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    f32x4 xi, vi;
                    const int it = i - k - 1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { xi[q] = L.Xs[(16 * i + g4 + q) * LDP + c16]; vi[q] = L.Vs[(16 * it + g4 + q) * LDP + c16]; }
                    sp = xty(vi, xi, sp);
                }
            }
            st4(L.Part + wave * 256 + lane * 4, sp);
            __syncthreads();
            BCLK(2);
            // ---- Wm = Th S Th^T;  Yt_j = Th Xt_j - 0.5 Wm Vt_j  (in place over X) ----
            {
                const f32x4 s = ld4(L.Part + lane * 4) + ld4(L.Part + 256 + lane * 4) + ld4(L.Part + 512 + lane * 4) + ld4(L.Part + 768 + lane * 4);
                const f32x4 tmp = xty0(s, ThT);                   // S Th^T
                f32x4 hwm = xty0(ThT, tmp);                       // Th S Th^T
                hwm *= -0.5f;
                for (int j = ilo + wave; j < NT; j += 4) {
                    float* xp = L.Xs + (16 * j + c16) * LDP + g4;
                    f32x4 y = xty0(ThT, ld4(xp));
                    if (j > k) y = xty(hwm, ld4(L.Vs + (16 * (j - k - 1) + c16) * LDP + g4), y);
                    st4(xp, y);
                }
            }
            __syncthreads();
            BCLK(3);
            // ---- A_ij -= V_i Y_j^T + Y_i V_j^T ----
            for (int i = ilo + wave; i < NT; i += 4) {
                const bool vi_ok = i > k;
                const f32x4 nvt = vi_ok ? neg4(ld4(L.Vs + (16 * (i - k - 1) + c16) * LDP + g4)) : ZERO4;
                const f32x4 nyt = neg4(ld4(L.Xs + (16 * i + c16) * LDP + g4));
                float* arow = At + (size_t)i * NT * 256 + lane * 4;
                const int jlo = vi_ok ? ilo : k + 1;               // rows without V only meet the columns with V
                int j = jlo;
                for (; j + 1 < NT; j += 2) {
                    f32x4 a0 = ld4(arow + (size_t)j * 256), a1 = ld4(arow + (size_t)(j + 1) * 256);
                    if (vi_ok) {
                        a0 = xty(nvt, ld4(L.Xs + (16 * j + c16) * LDP + g4), a0);
                        a1 = xty(nvt, ld4(L.Xs + (16 * j + 16 + c16) * LDP + g4), a1);
                    }
                    if (j > k) a0 = xty(nyt, ld4(L.Vs + (16 * (j - k - 1) + c16) * LDP + g4), a0);
                    if (j + 1 > k) a1 = xty(nyt, ld4(L.Vs + (16 * (j - k) + c16) * LDP + g4), a1);
                    st4(arow + (size_t)j * 256, a0);
                    st4(arow + (size_t)(j + 1) * 256, a1);
                }
                if (j < NT) {
                    f32x4 a0 = ld4(arow + (size_t)j * 256);
                    if (vi_ok) a0 = xty(nvt, ld4(L.Xs + (16 * j + c16) * LDP + g4), a0);
                    if (j > k) a0 = xty(nyt, ld4(L.Vs + (16 * (j - k - 1) + c16) * LDP + g4), a0);
                    st4(arow + (size_t)j * 256, a0);
                }
            }
            BCLK(4);
        }
        __syncthreads();
    }
    if constexpr (BACK) {
        // ---- W[b] = 0.5 (M + M^T) row-major, alpha[b, c, :] = a_c / sv_c ----
        const int b = t.b0 + bl;
        if (t.grad) {
            float* Wb = t.a.W + (size_t)b * N * N;
            f32x4 iden;
#pragma unroll
            for (int q = 0; q < 4; ++q) iden[q] = (g4 + q == c16) ? 0.5f : 0.f;
            const int ntt = NT * (NT + 1) / 2;
            for (int s = wave; s < ntt; s += 4) {
                int i = 0, rem = s;
                while (rem >= NT - i) { rem -= NT - i; ++i; }
                const int j = i + rem;                                                     // i <= j
                const f32x4 a = ld4(At + ((size_t)i * NT + j) * 256 + lane * 4), bt = ld4(At + ((size_t)j * NT + i) * 256 + lane * 4);
                f32x4 w = xty0(bt, iden);                                                  // 0.5 M_ji^T
                w += 0.5f * a;
                // tile (i, j)[4g + q][c] = W[16i + 4g + q][16j + c] = W[16j + c][16i + 4g + q]: a 16-byte store per lane into row 16j + c ...
                const int row = 16 * j + c16, col = 16 * i + g4;
                if (row < N) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (col + q < N) Wb[(size_t)row * N + col + q] = w[q];
                }
                if (i != j) {                                                              // ... and its mirror, row 16i + 4g + q, columns 16j + c
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r2 = 16 * i + g4 + q, c2 = 16 * j + c16;
                        if (r2 < N && c2 < N) Wb[(size_t)r2 * N + c2] = w[q];
                    }
                }
            }
        }
        for (int u = wave; u < NT * CP; u += 4) {
            const int i = u / CP, p = u - i * CP;
            const int cls = 16 * p + c16;
            if (cls < G.C) {
                const f32x4 a = ld4(Ut + (size_t)u * 256 + lane * 4);
                const float rs = 1.0f / t.a.sv[cls];
                float* al = t.a.alpha + ((size_t)b * G.C + cls) * N;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 16 * i + g4 + q;
                    if (r < N) al[r] = a[q] * rs;
                }
            }
        }
        // ---- the quadratic form against the ORIGINAL matrix.  The reduction's backward error (a few eps |E|) moves the small eigenvalues of K_c by a relative
        //      eps |E| sv / noise, which the quadratic form r^T K^-1 r feels in full (the log-determinant averages it out: measured 1e-6 against 3e-5 ... 9e-5 on
        //      class-correlated features, tools/band_mll_model.py).  With the residual rho = r - K alpha of the computed alpha, r^T K^-1 r = (r + rho)^T alpha up to
        //      second order: one product E a~ per episode (a~ = sv alpha, in A's tiles) restores the accuracy of a direct factorisation. ----
        {
            const brsrc Er = mk_rsrc(t.a.E + (size_t)b * N * N, (unsigned)((size_t)N * N * 4));
            const bool vec_ok = (N & 3) == 0;
            float qp[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};          // per class column: (2r - E a~ - mu a~).a~,  a~.E a~,  a~.a~,  sum a~
            for (int i = wave; i < NT; i += 4) {
                f32x4 ea[2] = {ZERO4, ZERO4};
                for (int j = 0; j < NT; ++j) {
                    // tile (j, i) of E in the accumulator layout: element [4g + q][c] = E[16j + 4g + q][16i + c] = E[16i + c][16j + 4g + q]
                    const int row = 16 * i + c16, col = 16 * j + g4;
                    f32x4 e;
                    if (vec_ok) {
                        e = bload4(Er, (row < N && col < N) ? (row * N + col) * 4 : OOB, 0);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            e[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Er, (row < N && col + q < N) ? (row * N + col + q) * 4 : OOB, 0, 0));
                    }
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        if (p < CP) ea[p] = xty(e, ld4(Ut + ((size_t)j * CP + p) * 256 + lane * 4), ea[p]);
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int cls = 16 * p + c16;
                    if (p < CP && cls < G.C) {
                        const f32x4 a = ld4(Ut + ((size_t)i * CP + p) * 256 + lane * 4);
                        const float mu = t.a.noise[cls] / t.a.sv[cls], mc = t.a.mean[cls];
                        const float* yc = t.a.Y + (size_t)b * t.a.y_bstride + (size_t)cls * N;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int r = 16 * i + g4 + q;
                            if (r < N) {
                                qp[p][0] += (2.0f * (yc[r] - mc) - ea[p][q] - mu * a[q]) * a[q];
                                qp[p][1] += ea[p][q] * a[q];
                                qp[p][2] += a[q] * a[q];
                                qp[p][3] += a[q];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    qp[p][v] += __shfl_xor(qp[p][v], 16, DKT_WAVE);
                    qp[p][v] += __shfl_xor(qp[p][v], 32, DKT_WAVE);
                }
            __syncthreads();
            if (lane < 16) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int v = 0; v < 4; ++v) L.Part[wave * 128 + v * 32 + 16 * p + lane] = qp[p][v];
            }
            __syncthreads();
            if (tid < G.C) {
                float sum[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) sum[v] = L.Part[v * 32 + tid] + L.Part[128 + v * 32 + tid] + L.Part[256 + v * 32 + tid] + L.Part[384 + v * 32 + tid];
                const size_t bc = (size_t)b * G.C + tid;
                const float sv = t.a.sv[tid], rs = 1.0f / sv, mu = t.a.noise[tid] * rs;
                const float quad = sum[0] * rs;
                if (t.grad) {
                    const float trz = t.a.dnoise[bc];                          // tr (B + mu)^-1 from the class kernel (NaN for a failed class)
                    t.a.dsv[bc] = 0.5f * (sum[1] * rs * rs - ((float)N - mu * trz) * rs);      // tr(M E), M = 0.5 (alpha alpha^T - K^-1):  alpha^T E alpha - tr(K^-1 E)
                    t.a.dnoise[bc] = 0.5f * (sum[2] * rs * rs - trz * rs);
                    t.a.dmean[bc] = sum[3] * rs;
                }
                t.a.logp[(size_t)b * G.C + tid] -= 0.5f * quad;            // the class kernel left -0.5 log det - N/2 log 2 pi there (NaN for a failed class)
            }
        }
    }
#ifdef DKT_BAND_CLOCKS
    if (tid == 0 && t.a.dnoise && bl < 64) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(t.ws + (size_t)t.bcnt * G.ep_floats) + (BACK ? 512 : 0) + bl * 8;
        for (int i = 0; i < 8; ++i) out[i] = clk[i];
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// Per class: the block LDL^T chain of B + mu I, one wave per class matrix
// ------------------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_class_kernel(BandArgs t) {
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

    f32x4 ngt_prev = ZERO4, yprev = ZERO4;
    float lsum = 0.f;
    int fail_at = 0;
    for (int j = 0; j < NT; ++j) {
        f32x4 P = ld4(At + ((size_t)j * NT + j) * 256 + lane * 4);
        float dmax = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (g4 + q == c16) P[q] = (16 * j + c16 < N) ? P[q] + mu : 1.0f;
        }
        if (j > 0) P = xty(ngt_prev, ld4(At + ((size_t)(j - 1) * NT + j) * 256 + lane * 4), P);       // - G_{j-1} S_{j-1}^T
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
        f32x4 y = col0 ? ld4(Ut + ((size_t)j * CP + pu) * 256 + (4 * g4 + cu) * 4) : ZERO4;
        if (j > 0) y = xty(ngt_prev, yprev, y);                                          // y_j = u_j - G_{j-1} y_{j-1}
        const f32x4 z = xty0(Pinv, y);
        st4(Pg + (size_t)j * 256 + lane * 4, Pinv);
        if (col0) st4(Zv + j * 16 + g4, z);
        if (j + 1 < NT) {
            const f32x4 St = ld4(At + ((size_t)j * NT + j + 1) * 256 + lane * 4);
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
    float trz = 0.f;
    for (int j = NT - 1; j >= 0; --j) {
        const f32x4 Pinv = ld4(Pg + (size_t)j * 256 + lane * 4);
        const f32x4 z = col0 ? ld4(Zv + j * 16 + g4) : ZERO4;
        f32x4 a, zd;
        if (j == NT - 1) {
            a = z;
            zd = Pinv;
        } else {
            const f32x4 Gj = ld4(Gg + (size_t)j * 256 + lane * 4);
            a = xty(neg4(Gj), a_next, z);
            zd = xty(Gj, xty0(zd_next, Gj), Pinv);
        }
        st4(Zg + (size_t)j * 256 + lane * 4, zd);
        if (col0) {
            st4(Am + ((size_t)j * CP + pu) * 256 + (4 * g4 + cu) * 4, a);
            float* at = AmT + ((size_t)pu * NT + j) * 256 + 64 * (cu >> 2) + (cu & 3);
#pragma unroll
            for (int q = 0; q < 4; ++q) at[4 * (g4 + q)] = a[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) trz += (g4 + q == c16 && 16 * j + c16 < N) ? zd[q] : 0.f;
        a_next = a;
        zd_next = zd;
    }
    lsum = wave_reduce_dpp<false>(lsum);
    trz = wave_reduce_dpp<false>(trz);
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
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// M = sum_c 0.5 cw_c (a_c a_c^T / sv_c - Z^c): a wave owns block columns, accumulates them over the classes in registers
// ------------------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void band_chain_kernel(BandArgs t) {
    const BandGeo& G = t.g;
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, g4 = (lane >> 2) & 12;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = G.NT, C = G.C, CP = G.CP;
    const int bl = blockIdx.x;
    const int nw = 4 * gridDim.y, wv = blockIdx.y * 4 + wave;
    float* ep = t.ws + (size_t)bl * G.ep_floats;
    float* At = ep + G.oA;
    const float* Gg = ep + G.oG;
    const float* Zg = ep + G.oZd;
    const float* AmT = ep + G.oAmT;
    const int b = t.b0 + bl;
    const int32_t* info = t.a.info + (size_t)b * C;
    f32x4 iden;
#pragma unroll
    for (int q = 0; q < 4; ++q) iden[q] = (g4 + q == c16) ? 1.0f : 0.f;
    // weights of the rank-C term by the class rows of an A^T tile: 0.5 cw_c / sv_c (0 beyond C and for a failed class)
    f32x4 ka[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 16 * p + g4 + q;
            ka[p][q] = (c < C && info[c < C ? c : 0] == 0) ? 0.5f * (t.a.cls_weight ? t.a.cls_weight[c] : 1.0f) / t.a.sv[c] : 0.f;
        }
    // zig-zag assignment of the block columns: wv, 2 nw - 1 - wv, 2 nw + wv, ...
    for (int base = 0; base < NT; base += 2 * nw) {
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int i = half == 0 ? base + wv : base + 2 * nw - 1 - wv;
            if (i >= NT) continue;
            f32x4 acc[BAND_MAXNT];
#pragma unroll
            for (int s = 0; s < BAND_MAXNT; ++s) acc[s] = ZERO4;
            for (int c = 0; c < C; ++c) {
                if (info[c] != 0) continue;
                const float wz = -0.5f * (t.a.cls_weight ? t.a.cls_weight[c] : 1.0f);
                f32x4 z = ld4(Zg + ((size_t)c * NT + i) * 256 + lane * 4);
                acc[0] += wz * z;
#pragma unroll
                for (int s = 1; s < BAND_MAXNT; ++s) {
                    if (s <= i) {
                        z = xty0(ld4(Gg + ((size_t)c * NT + i - s) * 256 + lane * 4), z);          // G_j^T z;  the sign alternates
                        acc[s] += ((s & 1) ? -wz : wz) * z;
                    }
                }
            }
            // rank-C term: M_ji += sum_p (A^T_pj)^T diag(ka) A^T_pi
            f32x4 ai[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                ai[p] = (p < CP) ? ld4(AmT + ((size_t)p * NT + i) * 256 + lane * 4) : ZERO4;
                ai[p] *= ka[p];
            }
#pragma unroll
            for (int s = 0; s < BAND_MAXNT; ++s) {
                if (s <= i) {
                    const int j = i - s;
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        if (p < CP) acc[s] = xty(ld4(AmT + ((size_t)p * NT + j) * 256 + lane * 4), ai[p], acc[s]);
                    st4(At + ((size_t)j * NT + i) * 256 + lane * 4, acc[s]);
                    if (s > 0) st4(At + ((size_t)i * NT + j) * 256 + lane * 4, xty0(acc[s], iden));
                }
            }
        }
    }
}

}  // namespace

bool dkt_mll_band_supports(int N, unsigned flags, int C) {
    if (flags & (DKT_MLL_WANT_CHOL | DKT_MLL_E_PER_CLASS | DKT_MLL_FORCE_GENERIC | DKT_MLL_FORCE_BLOCKED | DKT_MLL_FORCE_F32MFMA | DKT_MLL_FORCE_TILED)) return false;
    return N >= 128 && (N + 15) / 16 <= BAND_MAXNT && C >= 8 && C <= 32;
}

size_t dkt_mll_band_workspace_bytes(int B, int C, int N) {
    const BandGeo g = band_geo(N, C);
    const int bc = B < BAND_CHUNK ? B : BAND_CHUNK;
    size_t fl = (size_t)bc * g.ep_floats + 4096;
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
    const size_t lds = (size_t)ts_lds_floats(NT) * sizeof(float);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)band_twosided_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920) != hipSuccess) return DKT_ERR_LAUNCH;
        if (hipFuncSetAttribute((const void*)band_twosided_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920) != hipSuccess) return DKT_ERR_LAUNCH;
        attr_done = true;
    }
    const int Bc = a.B < BAND_CHUNK ? a.B : BAND_CHUNK;
    for (int b0 = 0; b0 < a.B; b0 += Bc) {
        const int bcnt = (a.B - b0 < Bc) ? a.B - b0 : Bc;
        t.b0 = b0;
        t.bcnt = bcnt;
        const int slots = NT * NT + 3 * NT * t.g.CP;
        hipLaunchKernelGGL(band_init_kernel, dim3((slots + 3) / 4, bcnt), dim3(256), 0, st, t);
        hipLaunchKernelGGL(band_twosided_kernel<false>, dim3(bcnt), dim3(256), lds, st, t);
        hipLaunchKernelGGL(band_class_kernel, dim3(bcnt, (a.C + 3) / 4), dim3(256), 0, st, t);
        if (t.grad) hipLaunchKernelGGL(band_chain_kernel, dim3(bcnt, 2), dim3(256), 0, st, t);
        hipLaunchKernelGGL(band_twosided_kernel<true>, dim3(bcnt), dim3(256), lds, st, t);
        MllArgs f = a;
        f.only_failed = a.info;
        dkt_mll_generic_global_launch(f, b0, bcnt, t.ws, st);
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
