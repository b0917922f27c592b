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
#include "dkt_mll.h"

namespace {

typedef __amdgpu_buffer_rsrc_t brsrc;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ brsrc mk_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(brsrc r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
__device__ __forceinline__ void bstore4(brsrc r, f32x4 x, int voff, int soff) {
    const u32x4 v = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}
__device__ __forceinline__ void bstore1(brsrc r, float x, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, 0);
}
#ifdef DKT_MFMA_CLOCKS      // measurement build (tools/mll_phase_clocks.py): s_memtime stamps per wave into the workspace pointer
#define DKT_CLK(i) do { __builtin_amdgcn_sched_barrier(0); clk[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DKT_CLK(i) do { } while (0)
#endif
constexpr int OOB = 0x7ffffff0;      // an offset every descriptor rejects: the load returns 0, the store is dropped

// D = C + X^T Y on accumulator-layout tiles
__device__ __forceinline__ f32x4 xty(const f32x4 x, const f32x4 y, f32x4 c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x[q], y[q], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x4 xty0(const f32x4 x, const f32x4 y) { return xty(x, y, (f32x4){0.f, 0.f, 0.f, 0.f}); }

template <int P>
__device__ __forceinline__ float rowbcast(float v) {          // DPP row_newbcast:P -- lane P of each 16-lane row to the whole row
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + P, 0xf, 0xf, false));
}

// accumulator register q (row group g holds row 4g + q) -> four registers holding rows q, 4+q, 8+q, 12+q in EVERY row group
__device__ __forceinline__ void spread_rows(float x, float& r0, float& r1, float& r2, float& r3) {
    const unsigned u = __float_as_uint(x);
    auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);    // [x0 x1 x0 x1], [x2 x3 x2 x3]
    auto lo = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);   // [x0 x0 x0 x0], [x1 x1 x1 x1]
    auto hi = __builtin_amdgcn_permlane16_swap(h[1], h[1], false, false);
    r0 = __uint_as_float(lo[0]); r1 = __uint_as_float(lo[1]); r2 = __uint_as_float(hi[0]); r3 = __uint_as_float(hi[1]);
}

struct Lane {
    int lane, g, c;
    bool g0, g1, g2;             // row-group predicates
};

// One pivot of the diagonal-tile sweep.  Replicated column layout, ONE register per row: lane c of x[i] is element [i][c];
// lanes c > p of a row i > p hold the (negated) Schur complement, lanes c <= p the rows of L^-1 under construction (the two
// halves obey the same update, so Gaussian elimination of [A | I] costs one DPP-fused FMA per row and pivot):
//      x[i] += row_newbcast_p(x[i]) * t,    t = x[p] / d  (c != p),   1/d - 1  (c == p: the multiplier column becomes L^-1's)
// Row p is final afterwards: scaled by 1/sqrt(d) it is row p of M = R^-T for c <= p (and of -R for c > p).
// A non-positive pivot is replaced by 1: the augmented pivot (whose raw value is the quadratic form), and the pivots of a
// matrix that fails (reported through dv, outputs poisoned by the caller).  Padding pivots are 1 by construction.
// x[i] += row_newbcast_P(x[i]) * t for i = P+1 .. 15 as ONE asm block of v_fmac_f32_dpp (hipcc does not fuse the DPP move into the
// FMA: it emits v_mov 0 / s_nop / v_mov_dpp / v_fmac per update).  The leading s_nop 1 covers the "VALU write -> DPP read: 2 wait
// states" hazard for whatever the compiler placed just before the block; inside it every instruction reads a register written
// at least one pivot earlier.
#define DKT_FMD(k) "v_fmac_f32_dpp %" #k ", %" #k ", %[t] row_newbcast:%[p] row_mask:0xf bank_mask:0xf\n\t"
template <int P>
__device__ __forceinline__ void sweep_rows(float (&x)[16], const float t) {
    if constexpr (P == 0)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) DKT_FMD(5) DKT_FMD(6) DKT_FMD(7) DKT_FMD(8) DKT_FMD(9) DKT_FMD(10) DKT_FMD(11) DKT_FMD(12) DKT_FMD(13) DKT_FMD(14)
                     : "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 1)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) DKT_FMD(5) DKT_FMD(6) DKT_FMD(7) DKT_FMD(8) DKT_FMD(9) DKT_FMD(10) DKT_FMD(11) DKT_FMD(12) DKT_FMD(13)
                     : "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 2)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) DKT_FMD(5) DKT_FMD(6) DKT_FMD(7) DKT_FMD(8) DKT_FMD(9) DKT_FMD(10) DKT_FMD(11) DKT_FMD(12)
                     : "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 3)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) DKT_FMD(5) DKT_FMD(6) DKT_FMD(7) DKT_FMD(8) DKT_FMD(9) DKT_FMD(10) DKT_FMD(11)
                     : "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 4)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) DKT_FMD(5) DKT_FMD(6) DKT_FMD(7) DKT_FMD(8) DKT_FMD(9) DKT_FMD(10)
                     : "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 5)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) DKT_FMD(5) DKT_FMD(6) DKT_FMD(7) DKT_FMD(8) DKT_FMD(9)
                     : "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 6)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) DKT_FMD(5) DKT_FMD(6) DKT_FMD(7) DKT_FMD(8)
                     : "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 7)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) DKT_FMD(5) DKT_FMD(6) DKT_FMD(7)
                     : "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 8)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) DKT_FMD(5) DKT_FMD(6)
                     : "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 9)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) DKT_FMD(5)
                     : "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 10)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4)
                     : "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 11)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3)
                     : "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 12)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1) DKT_FMD(2)
                     : "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 13)
        asm volatile("s_nop 1\n\t" DKT_FMD(0) DKT_FMD(1)
                     : "+v"(x[14]), "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (P == 14)
        asm volatile("s_nop 1\n\t" DKT_FMD(0)
                     : "+v"(x[15]) : [t] "v"(t), [p] "n"(P));
}
#undef DKT_FMD

template <int P>
__device__ __forceinline__ float pivot_bcast(const float xp) {
    float d;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(xp), "n"(P));
    return d;
}

template <int P>
__device__ __forceinline__ void sweep_pivot(float (&x)[16], float& dv, const Lane& ln) {
    float dneg = pivot_bcast<P>(x[P]);                    // -d_P, uniform
    const bool eq = ln.c == P;
    dv = eq ? -dneg : dv;
    dneg = (dneg < 0.f) ? dneg : -1.0f;
    const float rs = __builtin_amdgcn_rsqf(-dneg);        // 1 / sqrt(d)
    const float rs2 = rs * rs;
    const float t = eq ? (rs2 - 1.0f) : x[P] * rs2;
    x[P] = eq ? rs : x[P] * rs;
    sweep_rows<P>(x, t);
}

template <int P>
__device__ __forceinline__ void sweep_from(float (&x)[16], float& dv, const Lane& ln) {
    if constexpr (P < 16) {
        sweep_pivot<P>(x, dv, ln);
        sweep_from<P + 1>(x, dv, ln);
    }
}

// Diagonal tile S (accumulator layout, = -(Schur complement), symmetric) -> M = R^-T (accumulator layout).
// dv: lane c receives the raw pivot d_c;  x (out): the swept rows (CHOL: -R above the diagonal).
__device__ __forceinline__ f32x4 sweep_tile(const f32x4 S, float& dv, float (&x)[16], const Lane& ln) {
#pragma unroll
    for (int q = 0; q < 4; ++q) spread_rows(S[q], x[q], x[4 + q], x[8 + q], x[12 + q]);
    dv = 1.0f;
    sweep_from<0>(x, dv, ln);
    f32x4 M;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float v = ln.g0 ? x[q] : (ln.g1 ? x[4 + q] : (ln.g2 ? x[8 + q] : x[12 + q]));
        M[q] = (ln.c <= 4 * ln.g + q) ? v : 0.f;
    }
    return M;
}

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
    float nsv, dg, mc;       // -sv, -(noise + jitter), mean
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
            s[q] = (c16 == pN) ? (rok ? f.mc - yv[q] : 0.f) : s[q];
        }
        if constexpr (I == NT - 1) {
            const float yc = reinterpret_cast<const float*>(f.ys)[16 * I + c16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = s[q];
                v = (g4 + q == pN) ? ((c16 < pN) ? f.mc - yc : 0.f) : v;       // mirror row of the augmented column; pivot N = 0
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

// trailing update of block step 0 with the freshly formed tile as the C operand: S_ij = form(i, j) + R_0i^T R_0j
template <int NT, int I, int J>
__device__ __forceinline__ void form_trailing0(Tiles<NT>& T, const FormCtx& f) {
    if constexpr (I < NT) {
        T.t[I][J] = xty(T.t[0][I], T.t[0][J], form_tile<NT, I, J>(f));
        if constexpr (J + 1 < NT) form_trailing0<NT, I, J + 1>(T, f);
        else form_trailing0<NT, I + 1, I + 1>(T, f);
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
    Lane ln;
    ln.lane = tid & 63; ln.g = ln.lane >> 4; ln.c = ln.lane & 15;
    ln.g0 = ln.g == 0; ln.g1 = ln.g == 1; ln.g2 = ln.g == 2;
    const int C = a.C;
    const float qnan = __int_as_float(0x7fc00000);
    const int nrounds = (C + wpg - 1) / wpg;

#ifdef DKT_MFMA_CLOCKS
    unsigned long long clk[12] = {};
#endif
    DKT_CLK(0);
    for (int round = 0; round < nrounds; ++round) {
        // lane coordinates and sizes made opaque per round: keeps the compiler from hoisting (and then spilling) every mask and
        // address of the round body into the kernel prologue
        int c16 = ln.c, g4 = 4 * ln.g, lane = ln.lane, N = a.N;
        DKT_OPAQUE_V(c16); DKT_OPAQUE_V(g4); DKT_OPAQUE_V(lane); DKT_OPAQUE_S(N);
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
            f.pN = pN; f.c16 = c16; f.g4 = g4; f.lane = lane; f.nsv = -svc; f.mc = mc;
            int fail_at = 0;
            float jit = 0.f, lsum = 0.f, quad = 0.f;
            for (int attempt = 0; attempt <= a.max_tries; ++attempt) {
                jit = 0.f;
                if (attempt > 0) {
                    jit = a.jitter0;
                    for (int i = 1; i < attempt; ++i) jit *= 10.f;
                }
                f.dg = -(nzc + jit);
                fail_at = 0;
                lsum = 0.f;
                form_row0<NT, 0, 0>(T, f);
                // ---- phase 1: factorisation (block step 0 consumes the rest of E as it is read) ----
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    float dv, x[16];
                    __builtin_amdgcn_sched_barrier(0);       // keep block step k-1's updates above and step k's loads / MFMAs below the sweep
#ifdef DKT_MFMA_CLOCKS
                    const unsigned long long tsw0 = __builtin_amdgcn_s_memtime();
#endif
                    const f32x4 M = sweep_tile(T.t[k][k], dv, x, ln);
                    __builtin_amdgcn_sched_barrier(0);
#ifdef DKT_MFMA_CLOCKS
                    {
                        const unsigned long long tsw1 = __builtin_amdgcn_s_memtime();
                        clk[9] += tsw1 - tsw0;                      // all sweeps
                        if (k == 0) clk[10] = tsw1 - tsw0;
                        if (k == NT - 1) clk[11] = tsw1 - tsw0;
                    }
#endif
                    const bool valid = (k < NT - 1) || (c16 < pN);
                    const unsigned long long badm = __ballot(valid && !(dv > 0.f)) & 0xffffull;
                    const int first = (int)__builtin_ctzll(badm | 0x10000ull);
                    fail_at = (fail_at == 0 && badm != 0) ? 16 * k + first + 1 : fail_at;
                    lsum += (valid && ln.g0) ? __builtin_amdgcn_logf(dv) : 0.f;          // log2
                    if (k == NT - 1) quad = -__int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv), pN));
                    myst[k * 64 + lane] = M;
                    if constexpr (CHOL) {
                        // L[16k + c][16k + i] = R_kk[i][c] = -x[i] (i < c), 1 / M_kk[c][c] on the diagonal, zero above it
                        if (ln.g0 && valid) {
                            float* Lrow = a.L + bc * N * N + (size_t)(16 * k + c16) * N + 16 * k;
#pragma unroll
                            for (int i = 0; i < 16; ++i)
                                if (16 * k + i < N) Lrow[i] = (i < c16) ? -x[i] : ((i == c16) ? 1.0f / x[i] : 0.f);
                        }
                    }
                    if (k + 1 < NT) {
                        const f32x4 nV = xty0(M, negI);                                  // M^T (-I) = -V_kk
#pragma unroll
                        for (int j = k + 1; j < NT; ++j) T.t[k][j] = xty0(nV, T.t[k][j]);
                        if (k == 0) {
                            form_trailing0<NT, 1, 1>(T, f);
                        } else {
#pragma unroll
                            for (int i = k + 1; i < NT; ++i) {
#pragma unroll
                                for (int j = i; j < NT; ++j) T.t[i][j] = xty(T.t[k][i], T.t[k][j], T.t[i][j]);
                            }
                        }
                    }
                }
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
                        bstore4(Lr, T.t[k][j], row_ok ? ((16 * j + c16) * N + 16 * k + g4) * 4 : OOB, 0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool ok = (j < NT - 1) || (g4 + q < pN);
                            bstore1(Lr, 0.f, ok ? ((16 * k + c16) * N + 16 * j + g4 + q) * 4 : OOB, 0);
                        }
                    }
                }
            }
            // ---- phase 2: M = R^-T; M_ji (j > i) overwrites slot (i, j) ----
#pragma unroll
            for (int j = 1; j < NT; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 nV = xty0(myst[j * 64 + lane], negI);
#pragma unroll
                for (int i = 0; i < j; ++i) {
                    f32x4 Q = xty0(T.t[i][j], myst[i * 64 + lane]);
#pragma unroll
                    for (int k = i + 1; k < j; ++k) Q = xty(T.t[k][j], T.t[i][k], Q);
                    T.t[i][j] = xty0(nV, Q);
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
                    const float v = -(qn == 0 ? m[0] : qn == 1 ? m[1] : qn == 2 ? m[2] : m[3]);
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
#pragma unroll
                    for (int i = 0; i <= j; ++i) {           // i = 0 .. j-1, then the diagonal (which the others still read) last
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int k = j; k < NT; ++k) {
                            f32x4 A = (k == i) ? myst[k * 64 + lane] : T.t[i][k];
                            const f32x4 Bm = (k == j) ? myst[k * 64 + lane] : T.t[j][k];
                            if (k == NT - 1) A = A * sgn;
                            acc = xty(A, Bm, acc);
                        }
                        if (i < j) {
                            T.t[i][j] = acc;
                        } else {
                            // diagonal tile: M_jj is dead now, P''_jj takes its place in LDS; trace over the real rows
                            myst[j * 64 + lane] = acc;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const bool ok = (g4 + q == c16) && ((j < NT - 1) || (c16 < pN));
                                trpp += ok ? acc[q] : 0.f;
                            }
                        }
                    }
                }
            }
            DKT_CLK(5);
            lsum = wave_allsum(lsum);
            asum = wave_allsum(asum);
            trpp = wave_allsum(trpp);
            if (lane == 0) {
                const bool ok = fail_at == 0;
                a.logp[bc] = ok ? (-0.5f * quad - 0.34657359027997264f * lsum - (float)N * DKT_HALF_LOG_2PI) : qnan;
                a.jitter_used[bc] = jit;
                a.info[bc] = fail_at;
                if constexpr (GRAD) {
                    const float nz_eff = nzc + jit;
                    a.dmean[bc] = ok ? asum : qnan;
                    a.dnoise[bc] = ok ? -0.5f * trpp : qnan;                                    // 0.5 (alpha.alpha - tr K^-1)
                    a.dsv[bc] = ok ? 0.5f * ((quad - (float)N) + nz_eff * trpp) / svc : qnan;
                }
            }
            if constexpr (CHOL) {
                if (fail_at != 0) {
                    float* Lb = a.L + bc * N * N;
                    for (int idx = lane; idx < N * N; idx += 64) Lb[idx] = qnan;
                }
            }
            const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
            coef = (fail_at == 0) ? -0.5f * cw * svc : qnan;         // W_c = coef P''   (a failed class poisons W[b])
        }
        DKT_CLK(6);
        if constexpr (GRAD) {
            // ---- W[b] = sum over the classes of coef_c P''_c.  The staged E is dead once every wave is past its factorisation
            // (the barrier); its LDS becomes the exchange buffer. ----
            __syncthreads();
            DKT_OPAQUE_V(c16); DKT_OPAQUE_V(g4); DKT_OPAQUE_V(lane); DKT_OPAQUE_S(N);
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
    if (ln.lane == 0 && a.ws) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.ws) + ((size_t)blockIdx.x * (MFMA_MAX_WPG * EPW) + wall) * 12;
        for (int i = 0; i < 12; ++i) o[i] = clk[i];
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
    if (g && c && wpg == 5) hipLaunchKernelGGL((mll_mfma_kernel<NT, true, true, true>), grid, block, 0, st, a, wpg);
    else if (g && c) hipLaunchKernelGGL((mll_mfma_kernel<NT, true, true, false>), grid, block, 0, st, a, wpg);
    else if (g && wpg == 5) hipLaunchKernelGGL((mll_mfma_kernel<NT, true, false, true>), grid, block, 0, st, a, wpg);
    else if (g) hipLaunchKernelGGL((mll_mfma_kernel<NT, true, false, false>), grid, block, 0, st, a, wpg);
    else if (c) hipLaunchKernelGGL((mll_mfma_kernel<NT, false, true, false>), grid, block, 0, st, a, wpg);
    else hipLaunchKernelGGL((mll_mfma_kernel<NT, false, false, false>), grid, block, 0, st, a, wpg);
}

}  // namespace

bool dkt_mll_mfma_launch(const MllArgs& a, hipStream_t st) {
    const int nt = (a.N + 1 + 15) / 16;
    switch (nt) {
        case 1: launch_mfma<1>(a, st); return true;
        case 2: launch_mfma<2>(a, st); return true;
        case 3: launch_mfma<3>(a, st); return true;
        case 4: launch_mfma<4>(a, st); return true;
        case 5: launch_mfma<5>(a, st); return true;
        case 6: launch_mfma<6>(a, st); return true;
        case 7: launch_mfma<7>(a, st); return true;
        case 8: launch_mfma<8>(a, st); return true;
        default: return false;
    }
}
