// dkt_mll_tiled.hip -- exact-GP marginal likelihood for N + 1 > 128 (the 20-way shapes: N = 320 / 420, C = 20), the algorithm of
// dkt_mll_mfma.hip with the 16 x 16 tiles of every class matrix in a TILE ARRAY in memory instead of VGPRs.
//
// Replaces the same reference lines as dkt_mll_mfma.hip (methods/DKT.py:161-163, 177, 187, 252-254, 265, 330): GPyTorch's
// psd_safe_cholesky / inv_quad_logdet / cholesky_solve and their backward for the C one-vs-rest models K_c = sv_c E + noise_c I.
//
// Storage: per (episode, class) matrix the upper block triangle of the (N+1)-augmented, 16-padded matrix as NT (NT+1) / 2 tiles of
// 1 KB in the MFMA accumulator layout (lane l holds the 16 bytes at 16 l: element [4g + q][c], l = 16 g + c), tile rows contiguous.
// A tile is therefore ONE coalesced 1-KB load straight into MFMA operand registers -- no LDS staging, no transposes -- and every
// product is D += X^T Y = 4 x v_mfma_f32_16x16x4_f32 (dkt_mfma_tiles.h).  Three kernels, all LEFT-looking (every tile is written
// once; what a step needs from earlier steps is read back through L2 / MALL), 4 waves per workgroup, each wave a register block of
// 4 x MC accumulator tiles against 4 + MC operand tiles per step (double-buffered loads, no barriers in the K loops):
//   tiled_factor_kernel  (workgroup = matrix): block row I of 4 tile rows.  S_ij = form(E)_ij + sum_{k < i0} R_ki^T R_kj for the
//       wave's columns j = i0 + w (mod 4), then the 4 tile rows of the block one after the other: the diagonal tile is swept by
//       its owner (M_ii = R_ii^-T, pivots -> log det, first bad pivot, the quadratic form at the augmented pivot), M_ii goes
//       through LDS, every wave turns its tiles of the row into R_ij = (-V_ii)^T S_ij and applies them to the block's later rows.
//   tiled_invert_kernel  (workgroup = matrix): M = R^-T in place, block column J: slot (i, j) <- M_ji =
//       (-V_jj)^T sum_{i <= k < j} R_kj^T M_ki for the wave's rows i = w (mod 4); the columns of the block one after the other
//       (a column needs the finished ones of the same rows: registers of the same wave).  Row N of M is -alpha^T; the trace of
//       K^-1 - alpha alpha^T for the hyper-gradients is summed as the tiles become final; logp / gradients scalars are written here.
//   tiled_w_kernel       (workgroup = episode x block column of W): W_ij = sum_c coef_c sum_{k >= j} flip(M_ki)^T M_kj accumulated
//       over the classes IN REGISTERS and stored once (tile + mirror).
// The pass runs without jitter (attempt 0 of psd_safe_cholesky); an episode with a failed matrix is redone -- jitter ladder and all --
// by the generic kernel in a fix-up launch (dkt_mll.hip), exactly as the blocked path does.  No host read-back.
#include "dkt_h2_tiles.h"

namespace {

using namespace dkt_mfma;

constexpr int TB = 4;                         // tile rows / columns per block = waves per workgroup

struct TiledScal {                            // per matrix, factor -> invert -> w
    float lsum2, quad, coef;
    int msc, fail_at;
    float usc;                                // F16 pipeline: the power-of-two scale of the matrix' M = R^-T tiles in their f16-split storage (2^15 / bound)
    int erho;                                 // F16 pipeline: the augmented column (and row N of M, -alpha^T) carries an extra factor 2^-erho
    float trp, asp;                           // resident invert: trace / sum-of-alpha partial sums handed from one column range's launch to the next
};

struct TiledArgs {
    MllArgs a;
    float* tiles;                             // [nmat][NTT + 1][256]; tile NTT of every matrix is all zero
    float* etiles;                            // [episodes of the chunk][NTT + 1][256]: E[b] in tile layout (+ a zero tile)
    TiledScal* scal;                          // [nmat]
    int b0, NT, bcnt;
    int per_class;                            // DKT_MLL_E_PER_CLASS: E is [B,C,N,N] (etiles per MATRIX), W is [B,C,N,N] (the W kernel treats every matrix as an episode of one class)
};

__device__ __forceinline__ int tslot(const int NT, const int i, const int j) { return i * NT - (i * (i - 1)) / 2 + (j - i); }      // i <= j
__device__ __forceinline__ int toff(const int NT, const int i, const int j, const int lane) { return (tslot(NT, i, j) * 64 + lane) * 16; }
// tile (i, j) into operand registers: the slot offset is wave-uniform (SGPR operand of the buffer load), the lane part a constant VGPR
// (a tile that is structurally absent reads the matrix' all-zero tile, slot NT (NT+1) / 2, instead: the select is scalar)
__device__ __forceinline__ f32x4 tload(const brsrc Tr, const int NT, const bool ok, const int i, const int j, const int lane16) {
    return bload4(Tr, lane16, (ok ? tslot(NT, i, j) : (NT * (NT + 1)) / 2) * 1024);
}

struct Geo {
    int lane, c16, g4, pN, N, NT;
    Lane ln;
};
__device__ __forceinline__ Geo make_geo(const int tid, const int N, const int NT) {
    Geo g;
    g.lane = tid & 63; g.c16 = tid & 15; g.g4 = (tid >> 2) & 12; g.N = N; g.NT = NT;
    g.pN = N - 16 * (NT - 1);
    g.ln.lane = g.lane; g.ln.g = g.g4 >> 2; g.ln.c = g.c16;
    g.ln.g0 = g.ln.g == 0; g.ln.g1 = g.ln.g == 1; g.ln.g2 = g.ln.g == 2;
    return g;
}

struct FormRt {
    brsrc Et, yr;                 // the episode's E in tile layout (tiled_etile_kernel), this class' targets
    float nsv, dg, mc, rsc;
};

// E[b] -> tile layout, once per episode (shared by its C class matrices): tile (i, j), i <= j, element [4g+q][c] = E[16i + 4g + q][16j + c]
// = E[16j + c][16i + 4g + q] (symmetric: one 16-byte load per lane), zero beyond N; written as one coalesced 1-KB tile.
__global__ __launch_bounds__(256) void tiled_etile_kernel(const float* __restrict__ E, float* __restrict__ Et, long b0, int N, int NT) {     // b0: first base matrix of the chunk
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c16 = lane & 15, g4 = (lane >> 2) & 12;
    const int ntt = NT * (NT + 1) / 2, bl = blockIdx.y, slot = blockIdx.x * 4 + wave;
    if (slot > ntt) return;
    if (slot == ntt) {                                   // the all-zero tile absent tiles are read from
        reinterpret_cast<f32x4*>(Et + ((size_t)bl * (ntt + 1) + slot) * 256)[lane] = (f32x4){0.f, 0.f, 0.f, 0.f};
        return;
    }
    int i = 0, rem = slot;
    while (rem >= NT - i) { rem -= NT - i; ++i; }                           // slot -> (i, j): tile rows are contiguous
    const int j = i + rem;
    const brsrc Er = mk_rsrc(E + (size_t)(b0 + bl) * N * N, (unsigned)((size_t)N * N * 4));
    const int row = 16 * j + c16, col = 16 * i + g4;
    f32x4 e;
    if (col + 3 < N) {
        e = bload4(Er, row < N ? (row * N + col) * 4 : OOB, 0);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            e[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Er, (row < N && col + q < N) ? (row * N + col + q) * 4 : OOB, 0, 0));
    }
    reinterpret_cast<f32x4*>(Et + ((size_t)bl * (ntt + 1) + slot) * 256)[lane] = e;
}

// Tile (i, j), i <= j, of S = -K' / kappa in the accumulator layout from the episode's E tiles; the augmented column / row, its zero
// pivot and the identity padding as in form_tile of dkt_mll_mfma.hip.
__device__ __forceinline__ f32x4 form_from_e(const FormRt& f, const Geo& g, const int i, const int j, const f32x4 e) {
    const int NT = g.NT, N = g.N, pN = g.pN, c16 = g.c16, g4 = g.g4;
    f32x4 s;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = f.nsv * e[q];
        if (i == j) v = (g4 + q == c16) ? v + f.dg : v;
        s[q] = v;
    }
    if (j == NT - 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * i + g4 + q;
            const float yv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(f.yr, (c16 == pN && r < N) ? r * 4 : OOB, 0, 0));
            s[q] = (c16 == pN) ? ((r < N) ? (f.mc - yv) * f.rsc : 0.f) : s[q];
        }
        if (i == NT - 1) {
            const float yc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(f.yr, (c16 < pN) ? (16 * i + c16) * 4 : OOB, 0, 0));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = s[q];
                v = (g4 + q == pN) ? ((c16 < pN) ? (f.mc - yc) * f.rsc : 0.f) : v;
                v = (g4 + q > pN) ? ((g4 + q == c16) ? -1.0f : 0.f) : v;
                s[q] = v;
            }
        }
    }
    return s;
}

// c[u] += x[u]^T y (u = 0..3) resp. c[u] += x^T y[u]: four independent accumulator chains advanced together, so that no MFMA waits
// for the one issued just before it.
__device__ __forceinline__ void xty4_x(const f32x4 (&x)[TB], const f32x4 y, f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3) {
#ifdef DKT_TILED_NOMATH      // traffic-only measurement build (tools/tiled_traffic_ceiling.sh): the same tile stream, the products replaced by one add
    c0 += x[0] + y; c1 += x[1] + y; c2 += x[2] + y; c3 += x[3] + y;
    return;
#endif
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[0][q], y[q], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[1][q], y[q], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[2][q], y[q], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[3][q], y[q], c3, 0, 0, 0);
    }
}
__device__ __forceinline__ void xty4_y(const f32x4 x, const f32x4 (&y)[TB], f32x4 (&c)[TB]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int u = 0; u < TB; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q], y[u][q], c[u], 0, 0, 0);
    }
}

// the same on f16-split tiles (dkt_h2_tiles.h): 3 plane products per tile product, the four chains advanced together
__device__ __forceinline__ void xtyh4_x(const f32x4 (&x)[TB], const f32x4 y, f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3) {
    c0 = xtyh1<0>(x[0], y, c0); c1 = xtyh1<0>(x[1], y, c1); c2 = xtyh1<0>(x[2], y, c2); c3 = xtyh1<0>(x[3], y, c3);
    c0 = xtyh1<1>(x[0], y, c0); c1 = xtyh1<1>(x[1], y, c1); c2 = xtyh1<1>(x[2], y, c2); c3 = xtyh1<1>(x[3], y, c3);
    c0 = xtyh1<2>(x[0], y, c0); c1 = xtyh1<2>(x[1], y, c1); c2 = xtyh1<2>(x[2], y, c2); c3 = xtyh1<2>(x[3], y, c3);
}

// F16 pipeline (round 4; default with gradients, C <= 64): the tile arrays hold f16-SPLIT tiles (h, m planes of dkt_h2_tiles.h in the 16 bytes
// of a lane), so that every left-looking K loop multiplies them as they arrive -- 3 v_mfma_f32_16x16x16_f16 of 16 cycles per tile product instead
// of 4 v_mfma_f32_16x16x4_f32 of 32, no split arithmetic in the loops.  Scales: R tiles x 2^15 (|R_ij| <= 1 after the kappa scaling; the augmented
// column w = R^-T r is brought under the same bound by an extra factor 2^-erho on r, 2^erho >= sqrt(N / noise) max |r|, undone where the quadratic
// form, alpha and the trace leave the kernels); M tiles x usc = 2^15 / bound, bound = 2^msc / sqrt(noise) >= ||R^-T|| (a power of two per matrix:
// exact, invertible; the class weight is NOT folded in -- a class of weight zero still needs its M for alpha).  The last diagonal tile (it contains the
// unbounded row -alpha^T) stays fp32 until the invert kernel has read alpha off it.
__device__ __forceinline__ f32x4 neg_identity(const Geo& g) {
    f32x4 n;
#pragma unroll
    for (int q = 0; q < 4; ++q) n[q] = (g.g4 + q == g.c16) ? -1.0f : 0.0f;
    return n;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Factorisation.  Wave w owns the tile columns j = i0 + w + 4 bb of block row I.
// WGS = workgroups per CU the register budget is set for.  2: the operands of two K steps in registers (loads one step ahead).  3 (round 4, the F16
// pipeline's default; DKT_MLL_TILED_WGS=2 restores 2): ONE step's operands (168 VGPRs; the block rows after the first are instantiated with MC - 1 tile
// columns per wave, see dkt_mll_tiled_factor_row.inc) -- a lone workgroup of this kernel needs 147 us per N = 420
// matrix and two co-resident ones 165 us each: the kernel is bound by the serial chain of a block row (sweep -> barrier -> panel -> barrier -> updates),
// not by the matrix pipe or memory (profiles/r04/v4_factor_phase_clocks.log), so a third workgroup per CU fills idle pipes.
template <int MC, bool F16, int WGS>
__global__ __launch_bounds__(64 * TB, WGS) void tiled_factor_kernel(TiledArgs t) {
    __shared__ f32x4 mbuf[64];
    __shared__ f32x4 xbuf[TB][64];
    __shared__ float red[TB];
    __shared__ float lsum_w[TB];
    __shared__ int fail_w[TB];
    const MllArgs& a = t.a;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, NT = t.NT, C = a.C;
    // workgroup -> (episode, class): consecutive workgroup ids go to consecutive XCDs; the C classes of an episode read the same E[b],
    // so they are given ids 8 apart (one XCD, one L2)
    const int bl = ((int)(blockIdx.x >> 3) / C) * 8 + (int)(blockIdx.x & 7), c = (int)(blockIdx.x >> 3) % C;
    if (bl >= t.bcnt) return;
    const int m = bl * C + c, b = t.b0 + bl;
    const Geo g = make_geo(tid, N, NT);
    const int lane = g.lane, lane16 = g.lane * 16;
    const size_t ntt = (size_t)NT * (NT + 1) / 2;
    const brsrc Tr = mk_rsrc(t.tiles + (size_t)m * (ntt + 1) * 256, (unsigned)((ntt + 1) * 1024));
    const float* Eb = a.E + (t.per_class ? (size_t)b * C + c : (size_t)b) * N * N;
    // kappa = 4^msc >= max_i K_ii
    float emax = 0.f;
    for (int i = tid; i < N; i += 64 * TB) emax = fmaxf(emax, Eb[(size_t)i * (N + 1)]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) emax = fmaxf(emax, __shfl_xor(emax, o, DKT_WAVE));
    if (lane == 0) red[w] = emax;
    __syncthreads();
    emax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float svc = a.sv[c], mc = a.mean[c], nzc = a.noise[c];
    int ex;
    (void)frexpf(fmaf(svc, emax, nzc), &ex);
    const int msc = max(0, (ex + 1) >> 1);
    const float ikap = ldexpf(1.0f, -2 * msc);
    int erho = 0;
    float usc = 1.0f;
    if constexpr (F16) {
        // bound of the augmented column: |w|^2 = r^T K^-1 r <= N max|r|^2 / noise  ->  2^erho >= sqrt(N / noise) max|r|
        const float* yb = a.Y + (size_t)b * a.y_bstride + (size_t)c * N;
        float rmax = 0.f;
        for (int i = tid; i < N; i += 64 * TB) rmax = fmaxf(rmax, fabsf(yb[i] - mc));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rmax = fmaxf(rmax, __shfl_xor(rmax, o, DKT_WAVE));
        __syncthreads();
        if (lane == 0) red[w] = rmax;
        __syncthreads();
        rmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float nzp = nzc > 0.f ? nzc : 1e-6f;            // (a matrix with non-positive noise fails or not on its own; the bound only has to be finite)
        int ew;
        (void)frexpf(__builtin_sqrtf((float)N / nzp) * rmax, &ew);
        erho = min(max(ew, 0), 60);
        float uinv_unused;
        usc = scale_for(ldexpf(1.0f, msc) / __builtin_sqrtf(nzp), uinv_unused);
    }
    FormRt f;
    f.Et = mk_rsrc(t.etiles + (size_t)(t.per_class ? m : bl) * (ntt + 1) * 256, (unsigned)((ntt + 1) * 1024));
    f.yr = mk_rsrc(a.Y + (size_t)b * a.y_bstride + (size_t)c * N, (unsigned)(N * 4));
    f.nsv = -svc * ikap; f.dg = -nzc * ikap; f.mc = mc; f.rsc = ldexpf(1.0f, -msc - erho);
    const f32x4 negI = neg_identity(g);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    if (w == 0) bstore4(Tr, zero4, lane16, (int)ntt * 1024);               // the matrix' zero tile (read by this and the later kernels)
    __syncthreads();
    int fail_at = 0;
    float lsum = 0.f, quad = 0.f;
#ifdef DKT_TILED_CLOCKS      // measurement build (tools/tiled_phase_clocks.py): shader clocks per phase of this wave, summed over the block rows
    unsigned long long ck[6] = {0, 0, 0, 0, 0, 0}, c0 = __builtin_amdgcn_s_memtime();
#define TCLK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long c1 = __builtin_amdgcn_s_memtime(); ck[i] += c1 - c0; c0 = c1; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TCLK(i) do { } while (0)
#endif
    for (int i0 = 0; i0 < NT; i0 += TB) {
        if (WGS >= 3 && MC >= 2 && i0 > 0 && 4 * (MC - 1) >= NT - TB) {
#define MCB (MC >= 2 ? MC - 1 : 1)
#include "dkt_mll_tiled_factor_row.inc"
#undef MCB
        } else {
#define MCB MC
#include "dkt_mll_tiled_factor_row.inc"
#undef MCB
        }
    }
#ifdef DKT_TILED_CLOCKS
    TCLK(5);
    if (lane == 0 && w == 0 && a.dnoise) {                                  // phases of wave 0: form | K loop | own sweeps | waiting for a sweep | panel + updates | stores
        const size_t bcx = (size_t)b * C + c;
        a.logp[bcx] = (float)ck[0]; a.dsv[bcx] = (float)ck[1]; a.dmean[bcx] = (float)ck[2]; a.dnoise[bcx] = (float)ck[3];
        a.jitter_used[bcx] = (float)ck[4]; a.alpha[bcx * N] = (float)ck[5];
    }
#endif
    // ---- per-matrix scalars ----
    lsum = wave_allsum(lsum);
    if (lane == 0) { lsum_w[w] = lsum; fail_w[w] = fail_at; }
    if (w == ((NT - 1) & (TB - 1)) && lane == 0) red[0] = quad;            // the owner of the last diagonal tile
    __syncthreads();
    if (tid == 0) {
        int fa = 0;
        for (int k = 0; k < TB; ++k) fa = (fail_w[k] != 0 && (fa == 0 || fail_w[k] < fa)) ? fail_w[k] : fa;
        TiledScal s;
        s.lsum2 = lsum_w[0] + lsum_w[1] + lsum_w[2] + lsum_w[3] + (float)(2 * msc * N);
        s.quad = red[0] * ldexpf(1.0f, 2 * erho);
        s.coef = 0.f;
        s.msc = msc;
        s.fail_at = fa;
        s.usc = usc;
        s.erho = erho;
        s.trp = 0.f;
        s.asp = 0.f;
        t.scal[m] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// M = R^-T in place.  Wave w owns the tile rows i = w + 4 aa of block column J.
template <int MC, bool GRAD, bool F16, int WGS = 2>          // WGS: as tiled_factor_kernel (3: one K step's operands in registers, 3 workgroups per CU)
__global__ __launch_bounds__(64 * TB, WGS) void tiled_invert_kernel(TiledArgs t) {
    __shared__ float tr_w[TB], as_w[TB];
    __shared__ f32x4 dblk[10][64];               // the diagonal block of the current block column: R_kj (k < j, slot j (j-1) / 2 + k), M_jj (6 + j)
    const MllArgs& a = t.a;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, NT = t.NT, C = a.C;
    const int m = blockIdx.x, b = t.b0 + m / C, c = m % C;
    const Geo g = make_geo(tid, N, NT);
    const int lane = g.lane, lane16 = g.lane * 16, c16 = g.c16, g4 = g.g4, pN = g.pN;
    const size_t ntt = (size_t)NT * (NT + 1) / 2;
    const brsrc Tr = mk_rsrc(t.tiles + (size_t)m * (ntt + 1) * 256, (unsigned)((ntt + 1) * 1024));
    const size_t bc = (size_t)b * C + c;
    const brsrc ar = mk_rsrc(a.alpha + bc * N, (unsigned)(N * 4));
    const TiledScal sc = t.scal[m];
    const float rsc = ldexpf(1.0f, -sc.msc + (F16 ? sc.erho : 0));         // row N of M is -alpha^T 2^msc (2^-erho): undone here
    const float arow2 = F16 ? -ldexpf(1.0f, 2 * sc.erho) : -1.0f;          // weight of that row's squares in the trace
    const float usc = F16 ? sc.usc : 1.0f, uinv = 1.0f / usc;              // (a power of two)
    f32x4 keepv;                                                            // real rows of a tile of tile row NT - 1 (W must not see -alpha^T / the padding)
#pragma unroll
    for (int q = 0; q < 4; ++q) keepv[q] = (g.g4 + q < g.pN) ? 1.0f : 0.0f;
    const float qnan = __int_as_float(0x7fc00000);
    const bool failed = sc.fail_at != 0;
    const f32x4 negI = neg_identity(g);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool arow = (g.ln.g == (pN >> 2));
    const int qn = pN & 3;
    float trpp = 0.f, asum = 0.f;

    // trace of flip(M)^T M over the real columns, the part of one final tile: rows 16 j + 4g + q, columns 16 i + c
    auto trace_tile = [&](const f32x4 v, const int i, const int j) {
        const bool col_ok = 16 * i + c16 < N;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 16 * j + g4 + q;
            const float s = (row == N) ? arow2 : 1.0f;
            trpp += col_ok ? s * v[q] * v[q] : 0.f;
        }
    };
    auto alpha_tile = [&](const f32x4 v, const int i) {                    // tile of block row NT - 1: its row pN is -alpha^T (scaled)
        const float x = -(qn == 0 ? v[0] : qn == 1 ? v[1] : qn == 2 ? v[2] : v[3]) * rsc;
        const bool ok = arow && (16 * i + c16 < N);
        bstore1(ar, failed ? qnan : x, ok ? (16 * i + c16) * 4 : OOB, 0);
        asum += ok ? x : 0.f;
    };

    for (int j0 = 0; j0 < NT; j0 += TB) {
        f32x4 acc[MC][TB];
#pragma unroll
        for (int aa = 0; aa < MC; ++aa)
#pragma unroll
            for (int jj = 0; jj < TB; ++jj) acc[aa][jj] = zero4;
        // ---- K loop over the finished columns k < j0:  acc(i, j) += R_kj^T M_ki  (slot (i, k); the diagonal slot is M_ii) ----
        // (the K loop runs over k < j0 <= 4 (MC - 1), so its M operands are the rows i <= k < 4 (MC - 1): MC - 1 tiles per wave, whatever NT is)
        constexpr int MCY = MC > 1 ? MC - 1 : 1;
        auto loadk = [&](f32x4 (&X)[TB], f32x4 (&Y)[MCY], const int k) {
            const bool kin = k < j0;
#pragma unroll
            for (int jj = 0; jj < TB; ++jj) X[jj] = tload(Tr, NT, kin && j0 + jj < NT, k, j0 + jj, lane16);
#pragma unroll
            for (int aa = 0; aa < MCY; ++aa) {
                const int i = w + TB * aa;
                Y[aa] = tload(Tr, NT, kin && i <= k, i, k, lane16);
            }
        };
        auto mulk = [&](const f32x4 (&X)[TB], const f32x4 (&Y)[MCY], const int k) {
#pragma unroll
            for (int aa = 0; aa < MCY; ++aa) {
                if (w + TB * aa <= k) {
#ifdef DKT_TILED_NOMATH
#pragma unroll
                    for (int jj = 0; jj < TB; ++jj) acc[aa][jj] += X[jj] + Y[aa];
#else
                    if constexpr (F16) {
                        xtyh4_x(X, Y[aa], acc[aa][0], acc[aa][1], acc[aa][2], acc[aa][3]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int jj = 0; jj < TB; ++jj) acc[aa][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(X[jj][q], Y[aa][q], acc[aa][jj], 0, 0, 0);
                    }
#endif
                }
            }
        };
        // the 10 tiles of the diagonal block (all original R / M_jj), fetched by the waves together into LDS under the K loop's first loads
        f32x4 dl[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int tt = w + TB * u;                                      // 0..5: (kk, jj) = (0,1) (0,2) (1,2) (0,3) (1,3) (2,3); 6..9: the diagonal
            const int jj = tt >= 6 ? tt - 6 : (tt >= 3 ? 3 : (tt >= 1 ? 2 : 1));
            const int kk = tt >= 6 ? jj : tt - (jj * (jj - 1)) / 2;
            dl[u] = tload(Tr, NT, tt < 10 && j0 + jj < NT, j0 + kk, j0 + jj, lane16);
        }
        {
            f32x4 X0[TB], Y0[MCY], X1[TB], Y1[MCY];
            loadk(X0, Y0, 0);
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int tt = w + TB * u;
                if (tt < 10) {
                    f32x4 v = dl[u];
                    if constexpr (F16) {                                    // back to fp32 for the block's own columns: R tiles / 2^15, M_jj / usc (the last one is fp32)
                        if (tt < 6) v = join_h2(v) * TWOM15;
                        else if (j0 + (tt - 6) != NT - 1) v = join_h2(v) * uinv;
                    }
                    dblk[tt][lane] = v;
                }
            }
            if constexpr (WGS <= 2) {
                for (int k = 0; k < j0; k += 2) {
                    loadk(X1, Y1, k + 1);
                    mulk(X0, Y0, k);
                    loadk(X0, Y0, k + 2);
                    mulk(X1, Y1, k + 1);
                }
            } else {
                for (int k = 0; k < j0; ++k) {
                    mulk(X0, Y0, k);
                    loadk(X0, Y0, k + 1);
                }
            }
        }
        if constexpr (F16) {
            const float un = TWOM15 * uinv;                                 // the K loop accumulated (2^15 R)^T (usc M)
#pragma unroll
            for (int aa = 0; aa < MC; ++aa)
#pragma unroll
                for (int jj = 0; jj < TB; ++jj) acc[aa][jj] *= un;
        }
        __syncthreads();
        // ---- the block's columns, one after the other ----
#pragma unroll
        for (int jj = 0; jj < TB; ++jj) {
            const int j = j0 + jj;
            if (j < NT) {                                                   // uniform
#pragma unroll
                for (int kk = 0; kk < jj; ++kk) {
                    const int k = j0 + kk;
                    const f32x4 Xk = dblk[(jj * (jj - 1)) / 2 + kk][lane];                   // R_kj of the diagonal block (still original)
                    const f32x4 Mkk = dblk[6 + kk][lane];                                     // (row i = k belongs to wave kk)
#pragma unroll
                    for (int aa = 0; aa < MC; ++aa) {
                        const int i = w + TB * aa;
                        if (i <= k) acc[aa][jj] = xty(Xk, (i == k) ? Mkk : acc[aa][kk], acc[aa][jj]);
                    }
                }
                const f32x4 Mjj = dblk[6 + jj][lane];
                const f32x4 nV = xty0(Mjj, negI);
#pragma unroll
                for (int aa = 0; aa < MC; ++aa) {
                    const int i = w + TB * aa;
                    if (i < j) {
                        acc[aa][jj] = xty0(nV, acc[aa][jj]);
                        if constexpr (GRAD) trace_tile(acc[aa][jj], i, j);
                        if (j == NT - 1) alpha_tile(acc[aa][jj], i);
                    }
                }
                if (w == jj) {                                              // the diagonal tile of column j: trace / alpha once
                    if constexpr (GRAD) trace_tile(Mjj, j, j);
                    if (j == NT - 1) {
                        alpha_tile(Mjj, j);
                        if constexpr (F16) bstore4(Tr, split_h2(Mjj * keepv, usc), toff(NT, j, j, lane), 0);      // the last diagonal slot, now in W's format
                    }
                }
            }
        }
        __syncthreads();                                                    // every wave has read the block's R tiles
#pragma unroll
        for (int aa = 0; aa < MC; ++aa)
#pragma unroll
            for (int jj = 0; jj < TB; ++jj) {
                const int i = w + TB * aa, j = j0 + jj;
                f32x4 v = acc[aa][jj];
                if constexpr (F16) v = split_h2((j == NT - 1) ? v * keepv : v, usc);
                bstore4(Tr, v, (j < NT && i < j) ? toff(NT, i, j, lane) : OOB, 0);
            }
        __syncthreads();
    }
    trpp = wave_allsum(trpp);
    asum = wave_allsum(asum);
    if (lane == 0) { tr_w[w] = trpp; as_w[w] = asum; }
    __syncthreads();
    if (tid == 0) {
        const float tr = (tr_w[0] + tr_w[1] + tr_w[2] + tr_w[3]) * ldexpf(1.0f, -2 * sc.msc);
        const float as = as_w[0] + as_w[1] + as_w[2] + as_w[3];
        const float svc = a.sv[c], nzc = a.noise[c];
        const bool ok = !failed;
        a.logp[bc] = ok ? (-0.5f * sc.quad - 0.34657359027997264f * sc.lsum2 - (float)N * DKT_HALF_LOG_2PI) : qnan;
        a.jitter_used[bc] = 0.f;
        a.info[bc] = sc.fail_at;
        if constexpr (GRAD) {
            a.dmean[bc] = ok ? as : qnan;
            a.dnoise[bc] = ok ? -0.5f * tr : qnan;
            a.dsv[bc] = ok ? 0.5f * ((sc.quad - (float)N) + nzc * tr) / svc : qnan;
            const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
            t.scal[m].coef = ok ? -0.5f * cw * svc * ldexpf(1.0f, -2 * sc.msc) : qnan;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// W[b] = sum_c coef_c (K_c^-1 - alpha_c alpha_c^T)_s:  output block column J (grid x), episode (grid y); wave w owns rows i = w + 4 aa.
#ifndef DKT_TILED_W_WGS
#define DKT_TILED_W_WGS 2
#endif
// F16 (round 3, default; DKT_MLL_TILED_F16=0 keeps the fp32 products): the products run as scaled 2-way f16 splits on v_mfma_f32_16x16x16_f16
// (dkt_h2_tiles.h), 3 instructions of 16 cycles instead of 4 of 32 per tile product.  M_c = R_c^-T of the kappa-scaled matrix is bounded a
// priori, |M_c| <= ||K'_c^-1||^(1/2) <= 2^msc_c / sqrt(noise_c); the class weight is folded into the split scales,
//   A operand: sign(coef_c) g_c U M_ki,  B operand: g_c U M_kj,  g_c = sqrt |coef_c|,  U = 2^(15 - e), 2^e >= max_c g_c 2^msc_c / sqrt(noise_c)
// (one unit per episode), so the accumulators hold U^2 sum_c coef_c M_c^T M_c over the whole (class, k) stream.  The augmented row of M (-alpha^T,
// not covered by the bound) is zeroed in both operands; the rank-one terms -coef_c kappa_c alpha_c alpha_c^T are added on the VALU at the end from
// the alpha the invert kernel wrote.
template <int MC, bool F16, int WB>
__global__ __launch_bounds__(64 * WB, WB == 4 ? DKT_TILED_W_WGS : 2) void tiled_w_kernel(TiledArgs t) {
    __shared__ float qa_s[64], qb_s[64], ct_s[64];
    const MllArgs& a = t.a;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, NT = t.NT, C = a.C;
    // workgroup -> (episode, block column): the block columns of an episode read the same tile arrays class by class, so they are given
    // workgroup ids 8 apart (consecutive ids go to consecutive XCDs): one XCD, one L2
    const int nblk = (NT + WB - 1) / WB;
    const int bl = ((int)(blockIdx.x >> 3) / nblk) * 8 + (int)(blockIdx.x & 7);
    if (bl >= t.bcnt) return;
    const int j0 = ((int)(blockIdx.x >> 3) % nblk) * WB, b = t.b0 + bl;
    const Geo g = make_geo(tid, N, NT);
    const int lane = g.lane, lane16 = g.lane * 16, c16 = g.c16, g4 = g.g4, pN = g.pN;
    const size_t ntt = (size_t)NT * (NT + 1) / 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 sgn;
#pragma unroll
    for (int q = 0; q < 4; ++q) sgn[q] = (g4 + q == pN) ? (F16 ? 0.0f : -1.0f) : 1.0f;
    f32x4 acc[MC][WB];
#pragma unroll
    for (int aa = 0; aa < MC; ++aa)
#pragma unroll
        for (int jj = 0; jj < WB; ++jj) acc[aa][jj] = zero4;
    const int jmax = min(j0 + WB, NT) - 1;                                  // last column of the block
    // One descriptor over the C tile arrays of the episode; the (class, k) steps form ONE software-pipelined stream (the loads of
    // the next class's first step fly during the last step of the current one).
    const brsrc Tr = mk_rsrc(t.tiles + (size_t)bl * C * (ntt + 1) * 256, (unsigned)((size_t)C * (ntt + 1) * 1024));
    const TiledScal* sc = t.scal + (size_t)bl * C;
    const int nk = NT - j0, total = C * nk;
    float unit_inv2 = 1.0f;
    if constexpr (F16) {
        // per-class split scales and the episode's unit (see above); a failed class has coef = NaN and poisons the episode's W as in the fp32 kernel
        float bound = 0.f, coef = 0.f, g = 0.f;
        int msc = 0;
        if (tid < C) {
            coef = sc[tid].coef;
            msc = sc[tid].msc;
            g = __builtin_sqrtf(fabsf(coef));
            bound = g * ldexpf(1.0f, msc) / __builtin_sqrtf(a.noise[tid]);
            qa_s[tid] = bound;
        }
        __syncthreads();
        float mx = 1e-30f;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, qa_s[c]);               // (fmaxf drops the NaN of a failed class)
        __syncthreads();
        float inv;
        const float unit = scale_for(mx, inv);
        unit_inv2 = inv * inv;
        if (tid < C) {
            qa_s[tid] = (coef < 0.f ? -g : g) * unit + (coef - coef);      // (+ NaN for a failed class)
            qb_s[tid] = g * unit;
            ct_s[tid] = coef * ldexpf(1.0f, 2 * msc) * unit * unit;         // coefficient of alpha alpha^T in the accumulators' unit
        }
        __syncthreads();
    }
    // rows of this wave: aa = 0 is its row of the diagonal block (i = j0 + w, only the columns j >= i), aa >= 1 the rows above it
    // (i = j0 + w - WB aa >= 0, all WB columns)
    auto row_of = [&](const int aa) { return j0 + w - WB * aa; };
    int lc = 0, lk = j0;                                                    // load cursor
    auto loadk = [&](f32x4 (&A)[MC], f32x4 (&Bt)[WB]) {
        const int cbase = min(lc, C - 1) * (int)(ntt + 1), k = lk, zt = (int)ntt;
        const bool in = lc < C;
#pragma unroll
        for (int jj = 0; jj < WB; ++jj) {
            const int j = j0 + jj;
            Bt[jj] = bload4(Tr, lane16, (cbase + ((in && j <= k) ? tslot(NT, j, k) : zt)) * 1024);      // M_kj (the diagonal slot for k = j)
        }
#pragma unroll
        for (int aa = 0; aa < MC; ++aa) {
            const int i = row_of(aa);
            A[aa] = bload4(Tr, lane16, (cbase + ((in && i >= 0 && i <= jmax && i <= k) ? tslot(NT, i, k) : zt)) * 1024);
        }
        if (++lk == NT) { lk = j0; ++lc; }
    };
    int mcl = 0, mk = j0;                                                   // multiply cursor
    auto mulk = [&](f32x4 (&A)[MC], f32x4 (&Bt)[WB]) {
#ifdef DKT_TILED_NOMATH
#pragma unroll
        for (int aa = 0; aa < MC; ++aa)
#pragma unroll
            for (int u = 0; u < WB; ++u) acc[aa][u] += A[aa] + Bt[u];
        if (++mk == NT) { mk = j0; ++mcl; }
        return;
#endif
        if constexpr (F16) {
            const int cls = min(mcl, C - 1);
            const float qa = qa_s[cls], qb = qb_s[cls];
            const bool lastk = mk == NT - 1;
#pragma unroll
            for (int jj = 0; jj < WB; ++jj) Bt[jj] = split_h2(lastk ? Bt[jj] * sgn : Bt[jj], qb);
#pragma unroll
            for (int aa = 0; aa < MC; ++aa) A[aa] = split_h2(lastk ? A[aa] * sgn : A[aa], qa);
            if (j0 + w <= jmax) {
#pragma unroll
                for (int jj = 0; jj < WB; ++jj)
                    if (jj >= w) acc[0][jj] = xtyh(A[0], Bt[jj], acc[0][jj]);
            }
#pragma unroll
            for (int aa = 1; aa < MC; ++aa) {
                if (row_of(aa) >= 0) {
#pragma unroll
                    for (int u = 0; u < WB; ++u) acc[aa][u] = xtyh1<0>(A[aa], Bt[u], acc[aa][u]);
#pragma unroll
                    for (int u = 0; u < WB; ++u) acc[aa][u] = xtyh1<1>(A[aa], Bt[u], acc[aa][u]);
#pragma unroll
                    for (int u = 0; u < WB; ++u) acc[aa][u] = xtyh1<2>(A[aa], Bt[u], acc[aa][u]);
                }
            }
        } else {
            const float coef = sc[min(mcl, C - 1)].coef;
#pragma unroll
            for (int jj = 0; jj < WB; ++jj) Bt[jj] *= coef;
            if (mk == NT - 1) {
#pragma unroll
                for (int aa = 0; aa < MC; ++aa) A[aa] *= sgn;
            }
            if (j0 + w <= jmax) {                                           // the diagonal-block row: columns jj >= w
#pragma unroll
                for (int jj = 0; jj < WB; ++jj)
                    if (jj >= w) acc[0][jj] = xty(A[0], Bt[jj], acc[0][jj]);
            }
#pragma unroll
            for (int aa = 1; aa < MC; ++aa) {
                if (row_of(aa) >= 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int u = 0; u < WB; ++u) acc[aa][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[aa][q], Bt[u][q], acc[aa][u], 0, 0, 0);
                }
            }
        }
        if (++mk == NT) { mk = j0; ++mcl; }
    };
    {
        f32x4 A0[MC], B0[WB], A1[MC], B1[WB];
        loadk(A0, B0);
        for (int it = 0; it < total; it += 2) {
            loadk(A1, B1);
            mulk(A0, B0);
            loadk(A0, B0);
            if (it + 1 < total) mulk(A1, B1);
        }
    }
    if constexpr (F16) {
        // ---- the rank-one terms: acc(i, j) -= ct_c alpha_c[rows of i] alpha_c[columns of j]^T, class after class (alpha: NaN for a failed class) ----
#pragma unroll 2
        for (int c = 0; c < C; ++c) {
            const brsrc alr = mk_rsrc(a.alpha + ((size_t)b * C + c) * N, (unsigned)(N * 4));
            const float ct = ct_s[c];
            float aj[WB];
#pragma unroll
            for (int jj = 0; jj < WB; ++jj) {
                const int col = 16 * (j0 + jj) + c16;
                aj[jj] = ct * __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(alr, col < N ? col * 4 : OOB, 0, 0));
            }
#pragma unroll
            for (int aa = 0; aa < MC; ++aa) {
                const int i = row_of(aa);
                if (i >= 0 && i <= jmax) {                                  // uniform
                    f32x4 ai;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 16 * i + g4 + q;
                        ai[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(alr, row < N ? row * 4 : OOB, 0, 0));
                    }
#pragma unroll
                    for (int jj = 0; jj < WB; ++jj)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[aa][jj][q] = __builtin_fmaf(-ai[q], aj[jj], acc[aa][jj][q]);
                }
            }
        }
#pragma unroll
        for (int aa = 0; aa < MC; ++aa)
#pragma unroll
            for (int jj = 0; jj < WB; ++jj) acc[aa][jj] *= unit_inv2;
    }
    // ---- store: tile (i, j) and its mirror ----
    const brsrc Wr = mk_rsrc(a.W + (size_t)b * N * N, (unsigned)((size_t)N * N * 4));
#pragma unroll
    for (int aa = 0; aa < MC; ++aa)
#pragma unroll
        for (int jj = 0; jj < WB; ++jj) {
            const int i = j0 + w - WB * aa, j = j0 + jj;
            if (j < NT && i >= 0 && i <= j) {                               // uniform
                const f32x4 v = acc[aa][jj];
                const bool col_ok = 16 * j + c16 < N;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // a diagonal tile is stored from its upper half only (element and mirror image from the same register: W[b] is
                    // bitwise symmetric although the two halves of the accumulator were summed in different orders)
                    const bool row_ok = 16 * i + g4 + q < N && (i < j || g4 + q <= c16);
                    bstore1(Wr, v[q], (row_ok && col_ok) ? ((16 * i + g4 + q) * N + 16 * j + c16) * 4 : OOB, 0);
                    if (i == j) bstore1(Wr, v[q], (row_ok && col_ok && g4 + q < c16) ? ((16 * j + c16) * N + 16 * i + g4 + q) * 4 : OOB, 0);
                }
                if (i < j) bstore4(Wr, v, col_ok ? ((16 * j + c16) * N + 16 * i + g4) * 4 : OOB, 0);   // i < j <= NT - 1: 16 i + g4 + 3 < N
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------------------
// W with RESIDENT accumulators (round 4; default for C <= 64, DKT_MLL_TILED_WRES=0 restores the block-column kernel above).
// W[b] = sum_c coef_c M_c^T M_c - (rank-one terms) is a SYRK over the "strips" of M: strip k of class c = the tiles M_k0 .. M_kk (slots (i, k),
// i <= k), and W_ij += strip_k[i]^T strip_k[j] for every i <= j <= k.  The block-column kernel re-reads every strip once per block column of
// W (1229 tile reads per class matrix at NT = 27 for 378 distinct tiles; its traffic-only build, DKT_TILED_NOMATH, takes 4.75 of its 6.05 ms:
// profiles/r04/v0_tiled_traffic_ceiling.txt).  Here a workgroup of 4 waves keeps a COLUMN RANGE [c0, c1) of W -- a third of W at NT = 27 --
// in registers for the whole (class, k) stream, so a strip is read once per range that needs it (k >= c0, tiles i < c1: 664 tile reads per
// class matrix for the three ranges [0,16) [16,22) [22,27)), split ONCE while it is staged (fp32 tile -> registers -> scaled 2-way f16 split
// with the class weight folded in -> LDS), and every product takes both operands from that LDS image.
// Register blocking (v3; the first two versions loaded both operands per product: 14 scalar / vector / LDS instructions per MFMA, the matrix pipe
// 23 % busy, profiles/r04/v1_wres_pmc.txt): a wave owns CHUNKS of 4 tile rows x 2 tile columns, so a chunk's 8 products = 24 MFMAs take 6
// ds_read_b128 (two addresses + immediate offsets) and ~12 scalar instructions.  A chunk that straddles the diagonal also computes its tiles
// below the diagonal (never stored); their operands may lie beyond the strip (stale LDS): garbage stays in those accumulators.
// The strips are double-buffered in LDS, the global loads of step s + 2 fly while step s is multiplied; one barrier per step, and a step stages
// a GROUP of consecutive strips that fills a buffer (short strips share a step).
// Scales as in the block-column kernel (|M_c| <= 2^msc_c / sqrt(noise_c); U = one power-of-two unit per episode), but both operands now
// come from ONE image scaled by g_c U, g_c = sqrt|coef_c|, so the sign of coef_c cannot ride in an operand: the accumulators hold
// sigma U^2 sum_c coef_c M_c^T M_c and are negated when a class changes sigma (class weights of one sign: never).
struct WRanges { int ng; int c0[10]; };          // column ranges [c0[g], c0[g + 1]) of W, g < ng

// DMA (default; DKT_MLL_TILED_WDMA=0 restores the register-staged copy): the staged tiles go from memory straight into the LDS image with
// global_load_lds_dwordx4 (they are already in W's format) -- no staging registers, no copy phase; the 28 VGPRs that frees hold a second set of a chunk's
// operands, so the six ds_read_b128 of chunk u + 1 fly during the 24 MFMAs of chunk u.
typedef __attribute__((address_space(3))) unsigned char wres_lds_u8;
typedef __attribute__((address_space(1))) const unsigned char wres_glb_u8;
// NW = 8 (DKT_MLL_TILED_WNW): one 8-wave workgroup per CU instead of two of 4 waves -- twice the accumulators per workgroup, so wider column ranges
// (NT = 27: [0,22) [22,27) instead of three ranges, 488 instead of 664 tile reads per matrix; NT = 21: ONE range, 231 instead of 311).
template <int MAXC, bool DMA, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void tiled_wres_kernel(TiledArgs t, WRanges rg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wres_smem[];
    const MllArgs& a = t.a;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // DKT_MLL_E_PER_CLASS: every (episode, class) matrix has its own W -> a UNIT of this kernel is a matrix and its class "loop" has one entry
    // (C, b, bl below then count units: b indexes alpha / scal / W rows of the flat [B C] arrays)
    const int N = a.N, NT = t.NT, C = t.per_class ? 1 : a.C;
    const int ng = rg.ng;
    // workgroup -> (episode, range): the ranges of an episode read the same strips, so they get workgroup ids 8 apart (one XCD, one L2)
    const int bl = ((int)(blockIdx.x >> 3) / ng) * 8 + (int)(blockIdx.x & 7);
    if (bl >= (t.per_class ? t.bcnt * a.C : t.bcnt)) return;
    const int gi = (int)(blockIdx.x >> 3) % ng, b = (t.per_class ? t.b0 * a.C : t.b0) + bl;
    const int c0 = rg.c0[gi], c1 = rg.c0[gi + 1];
    const Geo g = make_geo(tid, N, NT);
    const int lane = g.lane, lane16 = g.lane * 16, c16 = g.c16, g4 = g.g4, pN = g.pN;
    const size_t ntt = (size_t)NT * (NT + 1) / 2;
    // LDS: [2][32] split tiles of 1 KB (28 slots for a step's group of strips + 4 that a diagonal chunk may read past it) | kap[C] | cr[C]
#ifndef DKT_WRES_NST8
#define DKT_WRES_NST8 6
#endif
    constexpr int NST = NW == 4 ? 7 : DKT_WRES_NST8, GT = NW * NST, BUFT = GT + 4;      // staged tiles per wave and step; tile slots per group; per buffer
    f32x4* sbuf = reinterpret_cast<f32x4*>(wres_smem);
    float* kap_s = reinterpret_cast<float*>(wres_smem + (size_t)2 * BUFT * 1024);
    float* cr_s = kap_s + 64;
    const brsrc Tr = mk_rsrc(t.tiles + (size_t)bl * C * (ntt + 1) * 256, (unsigned)((size_t)C * (ntt + 1) * 1024));
    const TiledScal* sc = t.scal + (size_t)bl * C;
    // The invert kernel left M_c in f16-split tiles scaled by usc_c (a power of two; the real rows only: the row -alpha^T and the padding are
    // zero), so a staged tile goes to LDS as it is and a product is (usc M)^T (usc M).  The class weight enters at the ACCUMULATORS: they hold
    // (sum_{c' <= c} kap_c' X_c') / kap_c with X_c = usc_c^2 M_c^T M_c and kap_c = coef_c / usc_c^2, rescaled by kap_c / kap_{c+1} when the class
    // changes (one multiply per accumulator register and class; a class of weight zero is skipped; a failed class has coef = NaN and poisons W).
    if (tid < C) {
        const float coef = sc[tid].coef, us = sc[tid].usc;
        kap_s[tid] = coef / (us * us);
        cr_s[tid] = coef * ldexpf(1.0f, 2 * sc[tid].msc);                   // coefficient of alpha alpha^T (alpha is unscaled)
    }
    __syncthreads();
    // this wave's chunks: the chunks of the range in column-pair-major order -- pair (j0, j0 + 1), j0 = c0, c0 + 2, ..., rows i0 = 0, 4, ... <= j0 + 1 --
    // dealt round-robin to the NW waves; chunk u of the wave = chunk number w + NW u.  Packed i0 | j0 << 8 in SGPRs for the whole kernel; a slot past
    // the end of the list gets j0 = 255 (it never takes part).
    int tc[MAXC];
    {
        int i0 = 0, j0 = c0;
        auto step1 = [&]() {                                                // next chunk in the enumeration
            i0 += 4;
            if (i0 > j0 + 1) { i0 = 0; j0 += 2; }
        };
        for (int x = 0; x < w; ++x) step1();
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            tc[u] = (j0 < c1) ? (i0 | (j0 << 8)) : (255 << 8);
            for (int x = 0; x < NW; ++x) step1();
        }
    }
    f32x4 acc[MAXC][8];                                                     // [chunk][4 rows x 2 columns: index 2 x + y]
#pragma unroll
    for (int u = 0; u < MAXC; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[u][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // A STEP stages a GROUP of consecutive strips [ka, kb) of one class -- as many as fit the GT tile slots of an LDS buffer (short strips
    // share a step), so that every step has enough products to cover the flight time of the next group's loads and there are fewer barriers.
    auto lim = [&](const int k) { return min(k + 1, c1); };                 // tiles of strip k this range needs
    auto group_end = [&](const int ka) {
        int cnt = 0, kb = ka;
        while (kb < NT && cnt + lim(kb) <= GT) { cnt += lim(kb); ++kb; }
        return kb;
    };
    int steps_per_class = 0;
    for (int ka = c0; ka < NT; ka = group_end(ka)) ++steps_per_class;
    const int total = C * steps_per_class;
    // The groups repeat class after class, so everything a step needs to know about its group is tabulated ONCE, in the lanes of a few VGPRs
    // (lane L = group L of a class; one v_readlane per use): gtab = ka | kb << 8, and per staged tile x of this wave (tile w + 4 x of the group)
    // stab[x] = tile slot of (strip k, tile i) in the matrix' tile array | last strip << 12 | valid << 13.  (The first versions walked the
    // strips with scalar loops in every step: 45 % of the kernel, profiles/r04/v2_wres_phase_clocks.log.)
    struct Grp { int c, gi; };
    int gtab, stab[NST];
    {
        int ka = c0, kb = group_end(c0);
        for (int tstep = 0; tstep < lane && tstep < steps_per_class; ++tstep) { ka = kb; kb = ka < NT ? group_end(ka) : ka; }
        const bool gvalid = lane < steps_per_class;
        gtab = ka | (kb << 8);
#pragma unroll
        for (int x = 0; x < NST; ++x) {
            int k = ka, i = w + NW * x;
            while (k < kb && i >= lim(k)) { i -= lim(k); ++k; }
            const bool ok = gvalid && k < kb;
            stab[x] = ok ? (tslot(NT, i, k) | ((k == NT - 1) ? 0x1000 : 0) | 0x2000) : 0;
        }
    }
    auto next_group = [&](const Grp gcur) {
        Grp n;
        n.c = gcur.c; n.gi = gcur.gi + 1;
        if (n.gi >= steps_per_class) { n.gi = 0; ++n.c; }
        return n;
    };
    auto load_step = [&](auto& S, const Grp gr) {
        const int cbase = min(gr.c, C - 1) * (int)(ntt + 1);
        const bool in = gr.c < C;
#pragma unroll
        for (int x = 0; x < NST; ++x) {
            const int e = __builtin_amdgcn_readlane(stab[x], gr.gi);
            S[x] = bload4(Tr, (in && (e & 0x2000)) ? lane16 : OOB, (cbase + (e & 0xfff)) * 1024);
        }
    };
    auto write_step = [&](const auto& S, const Grp gr, const int bufi) {
#pragma unroll
        for (int x = 0; x < NST; ++x) {
            const int e = __builtin_amdgcn_readlane(stab[x], gr.gi);
            if (e & 0x2000) sbuf[(bufi * BUFT + w + NW * x) * 64 + lane] = S[x];
        }
    };
    const unsigned char* tbase = reinterpret_cast<const unsigned char*>(t.tiles + (size_t)bl * C * (ntt + 1) * 256) + lane16;
    auto dma_step = [&](const Grp gr, const int bufi) {
        const int cbase = min(gr.c, C - 1) * (int)(ntt + 1);
        if (gr.c < C) {
#pragma unroll
            for (int x = 0; x < NST; ++x) {
                const int e = __builtin_amdgcn_readlane(stab[x], gr.gi);
                if (e & 0x2000)
                    __builtin_amdgcn_global_load_lds((wres_glb_u8*)(tbase + (size_t)(cbase + (e & 0xfff)) * 1024),
                                                     (wres_lds_u8*)(wres_smem + (bufi * BUFT + w + NW * x) * 1024), 16, 0, 0);
            }
        }
    };
    float curkap = 0.0f;                                                    // the accumulators' unit (0: nothing accumulated yet)
    bool skip = false;
    auto mul_step = [&](const Grp gr, const int bufi) {
        const int gk = __builtin_amdgcn_readlane(gtab, gr.gi), gka = gk & 255, gkb = gk >> 8;
        if (gr.gi == 0) {                                                   // a new class: bring the accumulators to its unit
            const float kc = kap_s[min(gr.c, C - 1)];
            skip = kc == 0.0f;
            if (!skip) {
                if (curkap != 0.0f) {
                    const float ratio = curkap / kc;
#pragma unroll
                    for (int u = 0; u < MAXC; ++u)
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[u][e] *= ratio;
                }
                curkap = kc;
            }
        }
        if (skip) return;
        const unsigned char* sb = reinterpret_cast<const unsigned char*>(sbuf) + bufi * BUFT * 1024 + lane16;
        int sbase = 0;
        for (int k = gka; k < gkb; ++k) {
            // column j takes part in strip k when j <= k; the chunks are column-pair-major, so the participants are a prefix of the wave's list
            const int kk = min(k, c1 - 1);
            if constexpr (DMA) {
                // two operand sets: chunk u + 1's reads are issued before chunk u's products
                f32x4 A[2][4], B0[2], B1[2];
                auto fetch = [&](const int set, int p) {
                    const unsigned char* pa = sb + (sbase + (p & 255)) * 1024;
                    const unsigned char* pb = sb + (sbase + (p >> 8)) * 1024;
#pragma unroll
                    for (int x = 0; x < 4; ++x) A[set][x] = *reinterpret_cast<const f32x4*>(pa + x * 1024);
                    B0[set] = *reinterpret_cast<const f32x4*>(pb);
                    B1[set] = *reinterpret_cast<const f32x4*>(pb + 1024);      // (stale when column j0 + 1 does not take part: not used then)
                };
                int p0 = tc[0];
                asm volatile("" : "+s"(p0));
                if ((p0 >> 8) <= kk) fetch(0, p0);
#pragma unroll
                for (int u = 0; u < MAXC; ++u) {
                    int p = tc[u];
                    asm volatile("" : "+s"(p));
                    const int j0 = p >> 8;
                    if (j0 <= kk) {
                        constexpr int cs = 0;
                        const int set = u & 1;
                        if (u + 1 < MAXC) {
                            // UNCONDITIONAL (a chunk that does not take part re-reads this chunk's tiles): a branch around the reads would join
                            // in front of the MFMAs and turn their s_waitcnt lgkmcnt(6) into lgkmcnt(0), i.e. no overlap
                            int pn = tc[u + 1 < MAXC ? u + 1 : u];
                            asm volatile("" : "+s"(pn));
                            fetch(set ^ 1, ((pn >> 8) <= kk) ? pn : p);
                        }
                        (void)cs;
                        if (j0 + 1 <= kk) {
#pragma unroll
                            for (int x = 0; x < 4; ++x) { acc[u][2 * x] = xtyh1<0>(A[set][x], B0[set], acc[u][2 * x]); acc[u][2 * x + 1] = xtyh1<0>(A[set][x], B1[set], acc[u][2 * x + 1]); }
#pragma unroll
                            for (int x = 0; x < 4; ++x) { acc[u][2 * x] = xtyh1<1>(A[set][x], B0[set], acc[u][2 * x]); acc[u][2 * x + 1] = xtyh1<1>(A[set][x], B1[set], acc[u][2 * x + 1]); }
#pragma unroll
                            for (int x = 0; x < 4; ++x) { acc[u][2 * x] = xtyh1<2>(A[set][x], B0[set], acc[u][2 * x]); acc[u][2 * x + 1] = xtyh1<2>(A[set][x], B1[set], acc[u][2 * x + 1]); }
                        } else {
#pragma unroll
                            for (int x = 0; x < 4; ++x) acc[u][2 * x] = xtyh1<0>(A[set][x], B0[set], acc[u][2 * x]);
#pragma unroll
                            for (int x = 0; x < 4; ++x) acc[u][2 * x] = xtyh1<1>(A[set][x], B0[set], acc[u][2 * x]);
#pragma unroll
                            for (int x = 0; x < 4; ++x) acc[u][2 * x] = xtyh1<2>(A[set][x], B0[set], acc[u][2 * x]);
                        }
                    }
                }
            } else {
#pragma unroll
            for (int u = 0; u < MAXC; ++u) {
                int p = tc[u];
                asm volatile("" : "+s"(p));                                 // (opaque: keeps the unpacked coordinates from being hoisted out of the loops -- and spilled)
                const int j0 = p >> 8;
                if (j0 <= kk) {
                    const unsigned char* pa = sb + (sbase + (p & 255)) * 1024;
                    const unsigned char* pb = sb + (sbase + j0) * 1024;
                    f32x4 A[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) A[x] = *reinterpret_cast<const f32x4*>(pa + x * 1024);
                    const f32x4 B0 = *reinterpret_cast<const f32x4*>(pb);
                    if (j0 + 1 <= kk) {
                        const f32x4 B1 = *reinterpret_cast<const f32x4*>(pb + 1024);
#pragma unroll
                        for (int x = 0; x < 4; ++x) { acc[u][2 * x] = xtyh1<0>(A[x], B0, acc[u][2 * x]); acc[u][2 * x + 1] = xtyh1<0>(A[x], B1, acc[u][2 * x + 1]); }
#pragma unroll
                        for (int x = 0; x < 4; ++x) { acc[u][2 * x] = xtyh1<1>(A[x], B0, acc[u][2 * x]); acc[u][2 * x + 1] = xtyh1<1>(A[x], B1, acc[u][2 * x + 1]); }
#pragma unroll
                        for (int x = 0; x < 4; ++x) { acc[u][2 * x] = xtyh1<2>(A[x], B0, acc[u][2 * x]); acc[u][2 * x + 1] = xtyh1<2>(A[x], B1, acc[u][2 * x + 1]); }
                    } else {
#pragma unroll
                        for (int x = 0; x < 4; ++x) acc[u][2 * x] = xtyh1<0>(A[x], B0, acc[u][2 * x]);
#pragma unroll
                        for (int x = 0; x < 4; ++x) acc[u][2 * x] = xtyh1<1>(A[x], B0, acc[u][2 * x]);
#pragma unroll
                        for (int x = 0; x < 4; ++x) acc[u][2 * x] = xtyh1<2>(A[x], B0, acc[u][2 * x]);
                    }
                }
            }
            }
            sbase += lim(k);
        }
    };
    {
        // one staging array: the loads of step s + 2 are issued right after step s + 1 went to LDS and fly during the barrier and step s + 1's products
        f32x4 S[DMA ? 1 : NST];
        Grp gA{0, 0};                                                       // the group being multiplied
        Grp gB = next_group(gA), gC = next_group(gB);                       // ... being split into LDS, ... being loaded
        if constexpr (DMA) {
            dma_step(gA, 0);
        } else {
            load_step(S, gA);
            write_step(S, gA, 0);
            load_step(S, gB);
        }
        __syncthreads();
#ifdef DKT_WRES_CLOCKS      // measurement build (tools/wres_phase_clocks.py): shader clocks of this wave per phase, summed over the steps
        unsigned long long ck[5] = {0, 0, 0, 0, 0}, tk0 = __builtin_amdgcn_s_memtime();
#define WCLK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tk1 = __builtin_amdgcn_s_memtime(); ck[i] += tk1 - tk0; tk0 = tk1; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WCLK(i) do { } while (0)
#endif
        for (int s = 0; s < total; ++s) {
            if constexpr (DMA) dma_step(gB, (s + 1) & 1);                   // step s + 1 lands in the other image while step s is multiplied
            mul_step(gA, s & 1);                                            // step s
#ifdef DKT_WRES_CLOCKS
            __builtin_amdgcn_s_waitcnt(0xc07f);                             // lgkmcnt(0): the step's LDS reads are back (its MFMAs may still run)
#endif
            WCLK(0);
#ifdef DKT_WRES_CLOCKS
            __builtin_amdgcn_s_waitcnt(0x0070 | 0x0f00);                    // vmcnt(0): how long the staged loads still needed
#endif
            WCLK(1);
            if constexpr (!DMA) write_step(S, gB, (s + 1) & 1);             // step s + 1 (that buffer was read last in step s - 1, a barrier ago)
            WCLK(2);
            if constexpr (!DMA) load_step(S, gC);                           // step s + 2
            gA = gB; gB = gC; gC = next_group(gC);
            WCLK(3);
            __syncthreads();
            WCLK(4);
        }
#ifdef DKT_WRES_CLOCKS
        if (lane == 0 && w == 0 && a.dnoise) {                              // wave 0 of range gi -> slot (b, gi) of the per-class outputs
            const size_t o = (size_t)b * C + gi;
            a.logp[o] = (float)ck[0]; a.dsv[o] = (float)ck[1]; a.dmean[o] = (float)ck[2]; a.dnoise[o] = (float)ck[3]; a.jitter_used[o] = (float)ck[4];
        }
#endif
    }
    // a tile of a chunk is real when it lies on or above the diagonal, inside the range
    auto real_tile = [&](const int i, const int j) { return i <= j && j < c1; };
    // ---- the rank-one terms: acc(i, j) -= coef_c kappa_c alpha_c[rows of i] alpha_c[columns of j]^T (alpha: NaN for a failed class), the alphas of
    //      a group of classes staged in the strip buffers ----
    {
        const int npad = 16 * NT;
        float* al_s = reinterpret_cast<float*>(wres_smem);
        const int cgrp = max(1, min(C, (2 * BUFT * 1024) / (npad * 4)));
#pragma unroll
        for (int u = 0; u < MAXC; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[u][e] *= curkap;                // -> sum_c coef_c M_c^T M_c (0 when every class weight is zero)
        for (int cb = 0; cb < C; cb += cgrp) {
            const int cn = min(cgrp, C - cb);
            __syncthreads();
            for (int idx = tid; idx < cn * npad; idx += 64 * NW) {
                const int c = idx / npad, r = idx - c * npad;
                al_s[idx] = (r < N) ? a.alpha[((size_t)b * C + cb + c) * N + r] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < MAXC; ++u) {
                const int i0 = tc[u] & 255, j0 = tc[u] >> 8;
                if (j0 < c1) {
                    for (int c = 0; c < cn; ++c) {
                        const float sct = cr_s[cb + c];
                        float aj[2];
#pragma unroll
                        for (int y = 0; y < 2; ++y) aj[y] = (j0 + y < NT) ? sct * al_s[c * npad + 16 * (j0 + y) + c16] : 0.f;
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            if (i0 + x < NT) {
                                const f32x4 ai = *reinterpret_cast<const f32x4*>(al_s + c * npad + 16 * (i0 + x) + g4);
#pragma unroll
                                for (int y = 0; y < 2; ++y)
#pragma unroll
                                    for (int q = 0; q < 4; ++q) acc[u][2 * x + y][q] = __builtin_fmaf(-ai[q], aj[y], acc[u][2 * x + y][q]);
                            }
                        }
                    }
                }
            }
        }
    }
    // ---- store: tile (i, j) and its mirror (as the block-column kernel: a diagonal tile from its upper half only -> bitwise symmetric) ----
    const brsrc Wr = mk_rsrc(a.W + (size_t)b * N * N, (unsigned)((size_t)N * N * 4));
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
        const int i0 = tc[u] & 255, j0 = tc[u] >> 8;
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int i = i0 + x, j = j0 + y;
                if (real_tile(i, j)) {                                      // uniform
                    const f32x4 v = acc[u][2 * x + y];
                    const bool col_ok = 16 * j + c16 < N;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool row_ok = 16 * i + g4 + q < N && (i < j || g4 + q <= c16);
                        bstore1(Wr, v[q], (row_ok && col_ok) ? ((16 * i + g4 + q) * N + 16 * j + c16) * 4 : OOB, 0);
                        if (i == j) bstore1(Wr, v[q], (row_ok && col_ok && g4 + q < c16) ? ((16 * j + c16) * N + 16 * i + g4 + q) * 4 : OOB, 0);
                    }
                    if (i < j) bstore4(Wr, v, col_ok ? ((16 * j + c16) * N + 16 * i + g4) * 4 : OOB, 0);   // i < j <= NT - 1: 16 i + g4 + 3 < N
                }
            }
    }
}
constexpr int WRES_MAXC = 5;                     // chunks (of 8 accumulator tiles) per wave: 160 VGPRs of the 256 a wave has at 2 workgroups per CU

// column ranges of W with at most 4 MAXC chunks each, cut at even distances from the start of a range (a chunk is a PAIR of columns):
// NT = 27 -> [0,16) [16,22) [22,27); NT = 21 -> [0,16) [16,21)
inline WRanges wres_ranges(const int NT, const int waves = 4) {
    WRanges r;
    r.ng = 0;
    r.c0[0] = 0;
    int cnt = 0;
    for (int j0 = 0; j0 < NT; j0 += 2) {                                    // j0 - c0 is even: the pairs restart with every range
        const int ch = (j0 + 1) / 4 + 1;                                    // chunks of the pair (j0, j0 + 1): rows i0 = 0, 4, ... <= j0 + 1
        if (cnt + ch > waves * WRES_MAXC) { r.c0[++r.ng] = j0; cnt = 0; }
        cnt += ch;
    }
    r.c0[++r.ng] = NT;
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// M = R^-T with RESIDENT accumulators (round 4; F16 pipeline only; opt-in with DKT_MLL_TILED_INVRES=1 -- parity-tested, NOT faster than the
// block-column tiled_invert_kernel it would replace: kept as the measured alternative).
// Slot (i, j), i < j, becomes M_ji = (-V_jj)^T sum_{i <= k < j} R_kj^T M_ki: per k a product of the R ROW k (tiles R_kj, j > k) with the STRIP k of M
// (tiles M_ki, i <= k) -- the structure of the W kernel with two operand images instead of one.  The block-column kernel re-reads the strips once per
// block column (1176 tile reads + 351 writes per matrix at NT = 27, at the ceiling of that access pattern: profiles/r04/v0_tiled_traffic_ceiling.txt);
// here a workgroup keeps a COLUMN RANGE of the slots in registers (the W kernel's ranges and 4 x 2 chunks), one launch per range, left to right:
// a strip k < c0 comes from memory (the earlier launches finished it), a strip k >= c0 is the range's own column k, finalised at the start of step k
// (scaled, multiplied by -V_kk^T on the fp32 pipe, counted into the trace / alpha, stored f16-split -- to memory for the W kernel and the later
// launches, and into the step's LDS image for this one).  767 tile reads + 378 writes per matrix.
// Step k: [finalise column k if k >= c0; barrier] products with image k; stage image k + 1 (registers -> LDS), issue the loads of image k + 2; barrier.
// Image k = slots 0 .. k: the strip (tile k = the diagonal slot's M_kk) | slots k + 1 ..: R_kj for j = max(k + 1, c0) .. c1 - 1 | slot 28: zeros.
template <int MAXC>
__global__ __launch_bounds__(256, 2) void tiled_invres_kernel(TiledArgs t, WRanges rg, int gi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wres_smem[];
    __shared__ float tr_w[TB], as_w[TB];
    const MllArgs& a = t.a;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = a.N, NT = t.NT, C = a.C;
    const int m = blockIdx.x, b = t.b0 + m / C, c = m % C;
    const int c0 = rg.c0[gi], c1 = rg.c0[gi + 1];
    const bool lastrange = gi == rg.ng - 1;
    const Geo g = make_geo(tid, N, NT);
    const int lane = g.lane, lane16 = g.lane * 16, c16 = g.c16, g4 = g.g4, pN = g.pN;
    const size_t ntt = (size_t)NT * (NT + 1) / 2;
    constexpr int NST = 7, GT = 4 * NST, BUFT = GT + 4, ZT = GT;           // staged tiles per wave and step; image slots; per buffer; the zero tile
    f32x4* sbuf = reinterpret_cast<f32x4*>(wres_smem);
    const brsrc Tr = mk_rsrc(t.tiles + (size_t)m * (ntt + 1) * 256, (unsigned)((ntt + 1) * 1024));
    const size_t bc = (size_t)b * C + c;
    const brsrc ar = mk_rsrc(a.alpha + bc * N, (unsigned)(N * 4));
    const TiledScal sc = t.scal[m];
    const float rsc = ldexpf(1.0f, -sc.msc + sc.erho);                      // row N of M is -alpha^T 2^msc 2^-erho
    const float arow2 = -ldexpf(1.0f, 2 * sc.erho);
    const float usc = sc.usc, uinv = 1.0f / usc, un = TWOM15 * uinv;        // the products accumulate (2^15 R)^T (usc M)
    const float qnan = __int_as_float(0x7fc00000);
    const bool failed = sc.fail_at != 0;
    const bool arow = (g.ln.g == (pN >> 2));
    const int qn = pN & 3;
    float trpp = 0.f, asum = 0.f;
    auto trace_tile = [&](const f32x4 v, const int i, const int j) {       // as in tiled_invert_kernel
        const bool col_ok = 16 * i + c16 < N;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 16 * j + g4 + q;
            const float sgn = (row == N) ? arow2 : 1.0f;
            trpp += col_ok ? sgn * v[q] * v[q] : 0.f;
        }
    };
    auto alpha_tile = [&](const f32x4 v, const int i) {
        const float x = -(qn == 0 ? v[0] : qn == 1 ? v[1] : qn == 2 ? v[2] : v[3]) * rsc;
        const bool ok = arow && (16 * i + c16 < N);
        bstore1(ar, failed ? qnan : x, ok ? (16 * i + c16) * 4 : OOB, 0);
        asum += ok ? x : 0.f;
    };
    // chunk table: as the W kernel (pairs of columns from c0, rows i0 = 0, 4, ... <= j0 + 1; the tiles on / below the diagonal never receive a product)
    int tc[MAXC];
    {
        int i0 = 0, j0 = c0;
        auto step1 = [&]() {
            i0 += 4;
            if (i0 > j0 + 1) { i0 = 0; j0 += 2; }
        };
        for (int x = 0; x < w; ++x) step1();
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            tc[u] = (j0 < c1) ? (i0 | (j0 << 8)) : (255 << 8);
            step1(); step1(); step1(); step1();
        }
    }
    f32x4 acc[MAXC][8];
#pragma unroll
    for (int u = 0; u < MAXC; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[u][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // staging table (lane = step k): item w + 4 x of image k -> tile slot in the matrix' array | image slot << 12 | valid << 20
    int stab[NST];
    {
        const int k = lane;
        const int nstrip = (k < c0) ? k + 1 : ((k <= NT - 2) ? 1 : 0);       // strip tiles from memory: all of it, or just the diagonal slot's M_kk
        const int js = max(k + 1, c0), nrow = max(c1 - js, 0);
#pragma unroll
        for (int x = 0; x < NST; ++x) {
            const int it = w + 4 * x;
            int e = 0;
            if (k <= c1 - 1 && it < nstrip + nrow) {
                if (it < nstrip) {
                    const int i = (k < c0) ? it : k;
                    e = tslot(NT, i, k) | (i << 12) | (1 << 20);
                } else {
                    const int j = js + (it - nstrip);
                    e = tslot(NT, k, j) | ((k + 1 + (j - js)) << 12) | (1 << 20);
                }
            }
            stab[x] = e;
        }
    }
    auto load_step = [&](f32x4 (&S)[NST], const int k) {
#pragma unroll
        for (int x = 0; x < NST; ++x) {
            const int e = __builtin_amdgcn_readlane(stab[x], min(k, 63));
            S[x] = bload4(Tr, (k < c1 && (e >> 20)) ? lane16 : OOB, (e & 0xfff) * 1024);
        }
    };
    auto write_step = [&](const f32x4 (&S)[NST], const int k) {
#pragma unroll
        for (int x = 0; x < NST; ++x) {
            const int e = __builtin_amdgcn_readlane(stab[x], min(k, 63));
            if (k < c1 && (e >> 20)) sbuf[((k & 1) * BUFT + ((e >> 12) & 31)) * 64 + lane] = S[x];
        }
    };
    // column k of the range: its accumulators are complete -> final tiles
    auto finalize = [&](const int k) {
        const int cur = k & 1;
        int g4o = g4, c16o = c16;
        asm volatile("" : "+v"(g4o), "+v"(c16o));                          // (rebuilt per call: 8 registers that need not live across the products)
        f32x4 negI, keepv;
#pragma unroll
        for (int q = 0; q < 4; ++q) { negI[q] = (g4o + q == c16o) ? -1.0f : 0.0f; keepv[q] = (g4o + q < pN) ? 1.0f : 0.0f; }
        f32x4 Mkk;
        if (k == NT - 1) Mkk = bload4(Tr, lane16, tslot(NT, k, k) * 1024);                       // the last diagonal tile is fp32 (row -alpha^T)
        else Mkk = join_h2(sbuf[(cur * BUFT + k) * 64 + lane]) * uinv;
        const f32x4 nV = xty0(Mkk, negI);
        auto one = [&](const f32x4 av, int i) {
            asm volatile("" : "+s"(i));                                     // (opaque: per-tile predicates / addresses must not be hoisted out of the step loop -- 400 spills)
            if (i < k) {                                                    // uniform
                const f32x4 v = xty0(nV, av * un);
                trace_tile(v, i, k);
                if (k == NT - 1) alpha_tile(v, i);
                const f32x4 sp = split_h2((k == NT - 1) ? v * keepv : v, usc);
                bstore4(Tr, sp, toff(NT, i, k, lane), 0);
                sbuf[(cur * BUFT + i) * 64 + lane] = sp;
                __builtin_amdgcn_sched_barrier(0);                          // one tile after the other: interleaving 8 of them costs 100 VGPRs
            }
        };
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            int p = tc[u];
            asm volatile("" : "+s"(p));
            const int i0 = p & 255, j0 = p >> 8;
            // the chunk's tiles of column k, picked with selects (no control flow around the accumulators: branches that read them made the
            // register allocator shuffle the whole set through scratch at the loop header)
            const bool first = j0 == k, any = first || j0 + 1 == k;
            f32x4 col[4];
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int q = 0; q < 4; ++q) col[x][q] = first ? acc[u][2 * x][q] : acc[u][2 * x + 1][q];
            if (any) {
#pragma unroll
                for (int x = 0; x < 4; ++x) one(col[x], i0 + x);
            }
        }
        if (w == 0) {                                                       // the diagonal tile: trace / alpha once
            trace_tile(Mkk, k, k);
            if (k == NT - 1) {
                alpha_tile(Mkk, k);
                bstore4(Tr, split_h2(Mkk * keepv, usc), toff(NT, k, k, lane), 0);               // the last diagonal slot, now in W's format
            }
        }
    };
    auto mul_step = [&](const int k) {
        const unsigned char* sb = reinterpret_cast<const unsigned char*>(sbuf) + (k & 1) * BUFT * 1024 + lane16;
        const int js = max(k + 1, c0);
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            int p = tc[u];
            asm volatile("" : "+s"(p));
            const int j0 = p >> 8, i0 = p & 255;
            // the chunk takes part while its second column is open (j0 + 1 > k).  Its first column may already be final (j0 == k) and its second one
            // may lie outside the range: those accumulators are dead / never stored, so both columns are always multiplied (their row-image slots are
            // stale but inside the buffer)
            if (j0 + 1 > k && j0 < c1 && i0 <= k) {
                const unsigned char* pr = sb + (k + 1 - js + j0) * 1024;   // the row image starts at slot k + 1 with column js
                const f32x4 R0 = *reinterpret_cast<const f32x4*>(pr), R1 = *reinterpret_cast<const f32x4*>(pr + 1024);
#pragma unroll
                for (int hx = 0; hx < 4; hx += 2) {                         // two rows at a time: 16 operand registers instead of 24
                    const f32x4 Sa = *reinterpret_cast<const f32x4*>(sb + ((i0 + hx <= k) ? i0 + hx : ZT) * 1024);
                    const f32x4 Sb = *reinterpret_cast<const f32x4*>(sb + ((i0 + hx + 1 <= k) ? i0 + hx + 1 : ZT) * 1024);
                    f32x4& a00 = acc[u][2 * hx], &a01 = acc[u][2 * hx + 1], &a10 = acc[u][2 * hx + 2], &a11 = acc[u][2 * hx + 3];
                    a00 = xtyh1<0>(R0, Sa, a00); a01 = xtyh1<0>(R1, Sa, a01); a10 = xtyh1<0>(R0, Sb, a10); a11 = xtyh1<0>(R1, Sb, a11);
                    a00 = xtyh1<1>(R0, Sa, a00); a01 = xtyh1<1>(R1, Sa, a01); a10 = xtyh1<1>(R0, Sb, a10); a11 = xtyh1<1>(R1, Sb, a11);
                    a00 = xtyh1<2>(R0, Sa, a00); a01 = xtyh1<2>(R1, Sa, a01); a10 = xtyh1<2>(R0, Sb, a10); a11 = xtyh1<2>(R1, Sb, a11);
                }
            }
        }
    };
    {
        f32x4 S[NST];
        if (w < 2) sbuf[(w * BUFT + ZT) * 64 + lane] = (f32x4){0.f, 0.f, 0.f, 0.f};
        load_step(S, 0);
        write_step(S, 0);
        load_step(S, 1);
        __syncthreads();
        for (int k = 0; k < c1; ++k) {
            if (k >= c0) {
                finalize(k);
                __syncthreads();
            }
            if (k <= c1 - 2) mul_step(k);
            write_step(S, k + 1);
            load_step(S, k + 2);
            __syncthreads();
        }
    }
    // ---- scalars: partial sums handed on in the matrix' TiledScal; the last range writes the outputs (as tiled_invert_kernel) ----
    trpp = wave_allsum(trpp);
    asum = wave_allsum(asum);
    if (lane == 0) { tr_w[w] = trpp; as_w[w] = asum; }
    __syncthreads();
    if (tid == 0) {
        const float trs = sc.trp + (tr_w[0] + tr_w[1] + tr_w[2] + tr_w[3]);
        const float ass = sc.asp + (as_w[0] + as_w[1] + as_w[2] + as_w[3]);
        if (!lastrange) {
            t.scal[m].trp = trs;
            t.scal[m].asp = ass;
        } else {
            const float tr = trs * ldexpf(1.0f, -2 * sc.msc);
            const float svc = a.sv[c], nzc = a.noise[c];
            const bool ok = !failed;
            a.logp[bc] = ok ? (-0.5f * sc.quad - 0.34657359027997264f * sc.lsum2 - (float)N * DKT_HALF_LOG_2PI) : qnan;
            a.jitter_used[bc] = 0.f;
            a.info[bc] = sc.fail_at;
            a.dmean[bc] = ok ? ass : qnan;
            a.dnoise[bc] = ok ? -0.5f * tr : qnan;
            a.dsv[bc] = ok ? 0.5f * ((sc.quad - (float)N) + nzc * tr) / svc : qnan;
            const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
            t.scal[m].coef = ok ? -0.5f * cw * svc * ldexpf(1.0f, -2 * sc.msc) : qnan;
        }
    }
}

// every image of every range fits the GT = 28 slots of a buffer (strip + row tiles)
inline bool invres_fits(const WRanges& r, const int NT) {
    for (int gq = 0; gq < r.ng; ++gq) {
        const int c0 = r.c0[gq], c1 = r.c0[gq + 1];
        if (c1 - 1 > 63) return false;
        for (int k = 0; k < c1; ++k) {
            const int nstrip = (k < c0) ? k + 1 : ((k <= NT - 2) ? 1 : 0);
            const int js = k + 1 > c0 ? k + 1 : c0, nrow = c1 - js > 0 ? c1 - js : 0;
            if (nstrip + nrow > 28 || k + 1 + nrow > 28) return false;
        }
    }
    return true;
}

int g_tiled_wnw = -1;
inline int tiled_wnw() {                   // DKT_MLL_TILED_WNW: waves per workgroup of the W kernel (4: two workgroups per CU; 8: one, wider column ranges; 84: 8 for the first range, 4 for the rest; 0 / unset: by NT)
    if (g_tiled_wnw < 0) {
        const char* v = dkt_variant_env("DKT_MLL_TILED_WNW");
        const int w = v ? atoi(v) : 0;
        g_tiled_wnw = (w == 8 || w == 4 || w == 84) ? w : 0;
    }
    return g_tiled_wnw;
}
int g_tiled_wdma = -1;
inline bool tiled_wdma() {
    if (g_tiled_wdma < 0) {
        const char* v = dkt_variant_env("DKT_MLL_TILED_WDMA");
        g_tiled_wdma = (v && v[0] == '0') ? 0 : 1;
    }
    return g_tiled_wdma != 0;
}

int g_tiled_wgs = -1;
inline int tiled_wgs() {
    if (g_tiled_wgs < 0) {
        const char* v = dkt_variant_env("DKT_MLL_TILED_WGS");
        g_tiled_wgs = (v && v[0] == '2') ? 2 : 3;
    }
    return g_tiled_wgs;
}

int g_tiled_invres = -1;
inline bool tiled_invres() {
    if (g_tiled_invres < 0) {
        const char* v = dkt_variant_env("DKT_MLL_TILED_INVRES");                      // default OFF: measured at parity with the block-column kernel (7.3 vs 6.5 ms at
        g_tiled_invres = (v && v[0] == '1') ? 1 : 0;                           // N = 420, 4.0 vs 4.1 at N = 320: 65 short steps per matrix)
    }
    return g_tiled_invres != 0;
}

int g_tiled_wres = -1;
inline bool tiled_wres() {
    if (g_tiled_wres < 0) {
        const char* v = dkt_variant_env("DKT_MLL_TILED_WRES");
        g_tiled_wres = (v && v[0] == '0') ? 0 : 1;
    }
    return g_tiled_wres != 0;
}

int g_tiled_f16 = -1;
inline bool tiled_f16() {
    if (g_tiled_f16 < 0) {
        const char* v = dkt_variant_env("DKT_MLL_TILED_F16");
        g_tiled_f16 = (v && v[0] == '0') ? 0 : 1;
    }
    return g_tiled_f16 != 0;
}
inline int tiled_nt(int N) { return (N + 1 + 15) / 16; }
inline size_t tiled_ws_floats(int Bc, int C, int N) {
    const size_t nt = tiled_nt(N), ntt = nt * (nt + 1) / 2, nmat = (size_t)Bc * C;
    size_t fl = nmat * (ntt + 1) * 256 + (size_t)Bc * (ntt + 1) * 256 + nmat * (sizeof(TiledScal) / sizeof(float)) + 64;
    const size_t gen = dkt_mll_generic_global_floats(Bc, N);               // the fix-up pass works in the same region
    return fl > gen ? fl : gen;
}
constexpr int TILED_CHUNK_MAX = 1024;      // episodes per pass over the workspace (N = 420, C = 20: 7.7 GB of tiles): the workspace is sized for it
int g_tiled_chunk = -1;
inline int tiled_chunk_episodes() {        // DKT_MLL_TILED_CHUNK (measurement aid: <= 1024; a chunk whose tile arrays fit the 256-MB memory-side cache)
    if (g_tiled_chunk < 0) {
        const char* v = getenv("DKT_MLL_TILED_CHUNK");
        const int c = v ? atoi(v) : TILED_CHUNK_MAX;
        g_tiled_chunk = (c >= 1 && c <= TILED_CHUNK_MAX) ? c : TILED_CHUNK_MAX;
    }
    return g_tiled_chunk;
}

template <int MC>
void tiled_chunk(const TiledArgs& t, int bcnt, bool grad, hipStream_t st) {
    const int nmat = bcnt * t.a.C;
    const dim3 fgrid(8 * ((bcnt + 7) / 8) * t.a.C), blk(64 * TB);
    hipLaunchKernelGGL(tiled_etile_kernel, dim3((t.NT * (t.NT + 1) / 2 + 1 + 3) / 4, t.per_class ? nmat : bcnt), dim3(256), 0, st, t.a.E, t.etiles,
                       t.per_class ? (long)t.b0 * t.a.C : (long)t.b0, t.a.N, t.NT);
#if defined(DKT_TILED_NOMATH)
    constexpr bool SPLIT_OK = false;                                        // the traffic-only build instruments the fp32 kernels
#else
    constexpr bool SPLIT_OK = true;
#endif
    // F16 pipeline (default for a call with gradients and C <= 64): f16-split tile arrays end to end -- factor and invert K loops on the f16 pipe,
    // W with resident accumulators.  DKT_MLL_TILED_WRES=0 -> round 3's kernels (fp32 factor / invert, block-column W on f16 products of fp32
    // tiles); DKT_MLL_TILED_F16=0 -> round 2's all-fp32 kernels.  A forward-only call (no W) runs the fp32 factor / invert.
    // Per-class base matrices (DKT_MLL_E_PER_CLASS) with gradients exist in this pipeline only (W per matrix = the W kernel with one class per unit).
    const bool msplit = SPLIT_OK && grad && (t.per_class || (tiled_f16() && tiled_wres() && t.a.C <= 64));
    // (3 workgroups per CU: 3.97 vs 4.94 ms at N = 320; at N = 420 -- MC = 7 -- only with the block-row body instantiated with MC - 1 columns from the second
    //  block row on, dkt_mll_tiled_factor_row.inc: 6.46 vs 7.05 ms; with MC columns throughout 7.03.  Two K steps' operands at 3 workgroups, where they fit
    //  (MC - 1 <= 5): 4.2 vs 4.0 ms, not used)
    if (msplit && tiled_wgs() >= 3) hipLaunchKernelGGL((tiled_factor_kernel<MC, true, 3>), fgrid, blk, 0, st, t);
#ifdef DKT_TWINS
    else if (msplit) hipLaunchKernelGGL((tiled_factor_kernel<MC, true, 2>), fgrid, blk, 0, st, t);
#endif
    else hipLaunchKernelGGL((tiled_factor_kernel<MC, false, 2>), fgrid, blk, 0, st, t);
#ifdef DKT_TILED_CLOCKS
    return;
#endif
    if (!grad) {
        hipLaunchKernelGGL((tiled_invert_kernel<MC, false, false>), dim3(nmat), blk, 0, st, t);
        return;
    }
    if (msplit) {
        const WRanges rg = wres_ranges(t.NT);
        const size_t lds = (size_t)2 * 32 * 1024 + 2 * 64 * sizeof(float);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)tiled_wres_kernel<WRES_MAXC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 1024 + 1024);
#ifdef DKT_TWINS
            (void)hipFuncSetAttribute((const void*)tiled_wres_kernel<WRES_MAXC, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 1024 + 1024);
            (void)hipFuncSetAttribute((const void*)tiled_invres_kernel<WRES_MAXC>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 1024 + 1024);
#endif
            attr_set = true;
        }
#ifdef DKT_TWINS         // the resident invert (measured at parity) and the two-workgroups-per-CU build
        if (tiled_invres() && invres_fits(rg, t.NT) && !t.per_class) {
            for (int gq = 0; gq < rg.ng; ++gq) hipLaunchKernelGGL((tiled_invres_kernel<WRES_MAXC>), dim3(nmat), dim3(256), (size_t)2 * 32 * 1024, st, t, rg, gq);
        } else if (tiled_wgs() < 3) {
            hipLaunchKernelGGL((tiled_invert_kernel<MC, true, true>), dim3(nmat), blk, 0, st, t);
        } else
#endif
        hipLaunchKernelGGL((tiled_invert_kernel<MC, true, true, 3>), dim3(nmat), blk, 0, st, t);
        const int units = t.per_class ? nmat : bcnt;
        // 8 waves when that makes W ONE column range where 4 waves need two (17 <= NT <= 21, N = 256 .. 335): 231 instead of 311 tile reads per matrix at NT = 21,
        // W 2.6 -> 1.9 ms per 1024 N = 320 episodes.  Where 8 waves still need two ranges the single workgroup per CU loses (NT = 27: W 4.0 -> 5.9 ms: each step's
        // LDS-DMA is exposed at its barrier with no second workgroup to fill in).
        const WRanges rg8 = wres_ranges(t.NT, 8);
        if (tiled_wdma() && (tiled_wnw() == 8 || (tiled_wnw() == 0 && rg.ng > 1 && rg8.ng == 1))) {
            static bool attr8 = false;
            if (!attr8) {
                (void)hipFuncSetAttribute((const void*)tiled_wres_kernel<WRES_MAXC, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (8 * DKT_WRES_NST8 + 4) * 1024 + 1024);
                attr8 = true;
            }
            hipLaunchKernelGGL((tiled_wres_kernel<WRES_MAXC, true, 8>), dim3(8 * ((units + 7) / 8) * rg8.ng), dim3(512), (size_t)2 * (8 * DKT_WRES_NST8 + 4) * 1024 + 2 * 64 * sizeof(float), st, t, rg8);
            return;
        }
        // NT >= 23 (N >= 352: the 20-way episode of 420 rows): the wide first range [0, 22) in an 8-wave workgroup, the rest -- at most 20 chunks -- in a 4-wave
        // workgroup of its own launch: 488 instead of 664 tile reads per matrix at NT = 27, W 4.0 -> 3.5 ms, the marginal likelihood 16.5 -> 15.95 ms per 1024 episodes
        // (the narrow range is what made two 8-wave ranges slow: 2.5 chunks per wave behind every barrier).
        int tail_chunks = 0;
        for (int j0 = rg8.c0[1]; rg8.ng == 2 && j0 < t.NT; j0 += 2) tail_chunks += (j0 + 1) / 4 + 1;
        if (tiled_wdma() && rg8.ng == 2 && tail_chunks <= 4 * WRES_MAXC && (tiled_wnw() == 84 || tiled_wnw() == 0)) {
            static bool attr84 = false;
            if (!attr84) {
                (void)hipFuncSetAttribute((const void*)tiled_wres_kernel<WRES_MAXC, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (8 * DKT_WRES_NST8 + 4) * 1024 + 1024);
                attr84 = true;
            }
            WRanges ra, rb;
            ra.ng = 1; ra.c0[0] = 0; ra.c0[1] = rg8.c0[1];
            rb.ng = 1; rb.c0[0] = rg8.c0[1]; rb.c0[1] = t.NT;
            hipLaunchKernelGGL((tiled_wres_kernel<WRES_MAXC, true, 8>), dim3(8 * ((units + 7) / 8)), dim3(512), (size_t)2 * (8 * DKT_WRES_NST8 + 4) * 1024 + 2 * 64 * sizeof(float), st, t, ra);
            hipLaunchKernelGGL((tiled_wres_kernel<WRES_MAXC, true>), dim3(8 * ((units + 7) / 8)), dim3(256), lds, st, t, rb);
            return;
        }
#ifdef DKT_TWINS
        if (!tiled_wdma()) { hipLaunchKernelGGL((tiled_wres_kernel<WRES_MAXC, false>), dim3(8 * ((units + 7) / 8) * rg.ng), dim3(256), lds, st, t, rg); return; }
#endif
        hipLaunchKernelGGL((tiled_wres_kernel<WRES_MAXC, true>), dim3(8 * ((units + 7) / 8) * rg.ng), dim3(256), lds, st, t, rg);
        return;
    }
#ifdef DKT_TWINS             // rounds 2 - 3: fp32 tile arrays, block-column W (DKT_MLL_TILED_WRES=0 / DKT_MLL_TILED_F16=0; also what served C > 64 until round 5)
    if (t.per_class) return;                                                 // (unreachable outside the traffic-only measurement build)
    hipLaunchKernelGGL((tiled_invert_kernel<MC, true, false>), dim3(nmat), blk, 0, st, t);
    // Block-column W (rounds 2-3).  Block columns of 8 tile columns with 8 waves (WB = 8: 880 instead of 1232 tile reads per class matrix at
    // NT = 27) were measured in round 3 and are NOT used: 25.8 vs 22.2 ms per 1024 cfg4 episodes for the whole marginal likelihood.
    const dim3 wgrid(8 * ((bcnt + 7) / 8) * ((t.NT + TB - 1) / TB));
    if (tiled_f16() && t.a.C <= 64) hipLaunchKernelGGL((tiled_w_kernel<MC, true, 4>), wgrid, blk, 0, st, t);
    else hipLaunchKernelGGL((tiled_w_kernel<MC, false, 4>), wgrid, blk, 0, st, t);
#endif
}

}  // namespace

void dkt_mll_tiled_reload_env() { g_tiled_f16 = -1; g_tiled_chunk = -1; g_tiled_wres = -1; g_tiled_invres = -1; g_tiled_wgs = -1; g_tiled_wdma = -1; g_tiled_wnw = -1; }      // dkt_reload_env()

bool dkt_mll_tiled_supports(int N, unsigned flags, int C) {
#ifndef DKT_TWINS
    // the product pipeline (f16-split tile arrays, W with resident accumulators) folds the class weights into split scales for up to 64 classes; a call with
    // gradients for more shared-matrix classes takes the blocked path
    if ((flags & DKT_MLL_WANT_GRAD) && !(flags & DKT_MLL_E_PER_CLASS) && C > 64) return false;
#endif
    return N + 1 > 128 && tiled_nt(N) <= 4 * 7 && !(flags & DKT_MLL_WANT_CHOL);      // N <= 447 (N + 1 <= 448 = 28 tiles): 4 x 7 register tiles per wave without spills
}

// DKT_MLL_E_PER_CLASS: the base matrices take as many tiles as the factors, so a pass covers half as many episodes (and at most 65535 matrices: the
// conversion kernel's grid); the workspace of ANY call (the size does not depend on the flags) covers both forms.
inline int tiled_pc_chunk(int B, int C, int limit) {
    int bc = limit / 2 < 1 ? 1 : limit / 2;
    if (bc > 65535 / C) bc = 65535 / C;
    if (bc < 1) bc = 1;
    return B < bc ? B : bc;
}
inline size_t tiled_pc_ws_floats(int Bc, int C, int N) {
    const size_t nt = tiled_nt(N), ntt = nt * (nt + 1) / 2, nmat = (size_t)Bc * C;
    return 2 * nmat * (ntt + 1) * 256 + nmat * (sizeof(TiledScal) / sizeof(float)) + 64;
}
size_t dkt_mll_tiled_workspace_bytes(int B, int C, int N) {
    const int bc = B < TILED_CHUNK_MAX ? B : TILED_CHUNK_MAX;
    const size_t shared = tiled_ws_floats(bc, C, N), pc = C <= 65535 ? tiled_pc_ws_floats(tiled_pc_chunk(B, C, TILED_CHUNK_MAX), C, N) : 0;
    return (shared > pc ? shared : pc) * sizeof(float);
}
// the form one call needs (dkt_mll_workspace_bytes_for): shared base matrix or one per class
size_t dkt_mll_tiled_workspace_bytes_form(int B, int C, int N, bool per_class) {
    if (per_class) return C <= 65535 ? tiled_pc_ws_floats(tiled_pc_chunk(B, C, TILED_CHUNK_MAX), C, N) * sizeof(float) : 0;
    return tiled_ws_floats(B < TILED_CHUNK_MAX ? B : TILED_CHUNK_MAX, C, N) * sizeof(float);
}

// Returns 0 on success, a negative DKT status otherwise.
int dkt_mll_tiled_launch(const MllArgs& a, void* workspace, size_t ws_bytes, hipStream_t st) {
    const int N = a.N, C = a.C;
    const bool pc = (a.flags & DKT_MLL_E_PER_CLASS) != 0;
    if (!workspace || ws_bytes < dkt_mll_tiled_workspace_bytes_form(a.B, C, N, pc)) return DKT_ERR_WORKSPACE;
    const int NT = tiled_nt(N);
    if (pc && C > 65535) return DKT_ERR_TOO_LARGE;
    const int Bc = pc ? tiled_pc_chunk(a.B, C, tiled_chunk_episodes()) : (a.B < tiled_chunk_episodes() ? a.B : tiled_chunk_episodes());
    const size_t ntt = (size_t)NT * (NT + 1) / 2, nmat_max = (size_t)Bc * C;
    TiledArgs t;
    t.a = a;
    t.per_class = pc ? 1 : 0;
    t.tiles = (float*)workspace;
    t.etiles = t.tiles + nmat_max * (ntt + 1) * 256;
    t.scal = (TiledScal*)(t.etiles + (pc ? nmat_max : (size_t)Bc) * (ntt + 1) * 256);
    t.NT = NT;
    const bool grad = (a.flags & DKT_MLL_WANT_GRAD) != 0;
    const int mc = (NT + TB - 1) / TB;
    for (int b0 = 0; b0 < a.B; b0 += Bc) {
        const int bcnt = (a.B - b0 < Bc) ? a.B - b0 : Bc;
        t.b0 = b0;
        t.bcnt = bcnt;
        switch (mc) {
            case 3: tiled_chunk<3>(t, bcnt, grad, st); break;
            case 4: tiled_chunk<4>(t, bcnt, grad, st); break;
            case 5: tiled_chunk<5>(t, bcnt, grad, st); break;
            case 6: tiled_chunk<6>(t, bcnt, grad, st); break;
            case 7: tiled_chunk<7>(t, bcnt, grad, st); break;
            default: return DKT_ERR_BAD_ARG;
        }
        // fix-up: episodes with a failed matrix are redone by the generic kernel with the jitter ladder (its global working
        // matrices reuse the tile region, which is dead by now)
#if !defined(DKT_TILED_CLOCKS) && !defined(DKT_TILED_NOMATH)
        {
            // shared E: the generic kernel redoes the failed EPISODES (W[b] is a sum over the classes); per-class base matrices: the failed
            // MATRICES, one workgroup each (E[b, c] -> W[b, c]).  Its (N + 1) x (N | 1) working matrices fit the dead tile + E-tile regions:
            // 2 (NTT + 1) 256 = 256 NT^2 + 256 NT + 512 >= (N + 1)^2 floats per matrix (per-class), C (NTT + 1) 256 per episode (shared).
            MllArgs f = a;
            f.only_failed = a.info;
            dkt_mll_generic_global_launch(f, b0, bcnt, t.tiles, st);
        }
#endif
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
