// dkt_mll_reg.hip -- register-resident exact-GP marginal likelihood for N + 1 <= 128 (the few-shot
// shapes: N = 105, 85, 25, 19, 5).
//
// One 256-thread workgroup per episode, classes in sequence.  The (N+1) x N working matrix of
// dkt_mll.hip (L below the diagonal, U = L^-T above it, w = L^-1 r in row N) never touches LDS: it is
// distributed 2-D cyclically over the 16 x 16 thread grid,
//       thread (ty, tx) owns  Mw[ty + 16*pi][tx + 16*ji],  pi, ji in [0, NT),  NT = ceil((N+1)/16),
// i.e. NT*NT registers per lane (49 for N = 105), rows held in pairs (pi = 2m, 2m+1) so the rank-1 update runs on
// v_pk_fma_f32.  A sweep step costs ONE barrier and ONE LDS round trip: the 16 owners of column k publish it (raw) to a
// double-buffered LDS vector; every thread reads the pivot and its NT row / NT column entries, takes v_rcp_f32, updates
// block column KQ first, the owners of column k+1 publish it at once, and the remaining block columns are updated off
// the critical path.  KQ = k / 16 is a template parameter, so every register index is static and blocks that are
// structurally untouched (ji < KQ, or KQ < pi < ji) cost nothing.  Columns stay unscaled during the sweep; pivots,
// column scales 1 / L_kk, log det and the first non-positive pivot are all read off the diagonal slots afterwards.
//
// Gradient: K^-1 = U U^T is a Gram matrix of the rows of U.  Per class, U goes through LDS one 16-column chunk at a
// time and each wave accumulates its tile rows of U U^T on v_mfma_f32_16x16x4_f32 (four per-wave instantiations, each
// accumulator first touched when its row block becomes non-zero); the epilogue forms coef_c (alpha alpha^T - U U^T)
// with alpha from LDS and adds it into W[b] in memory, class after class (same lane, same word: deterministic).
// The per-class hyper gradients need only scalars:  tr K^-1 = |U|_F^2,  alpha.alpha,  1.alpha,  r.alpha:
//     dnoise = 0.5 (alpha.alpha - tr K^-1)
//     dsv    = 0.5 ((r.alpha - N) - (noise + jitter) (alpha.alpha - tr K^-1)) / sv      [sv E = K - (noise+jitter) I]
//
// Replaces the same reference lines as dkt_mll.hip (methods/DKT.py:161-163,177,187,252-254,265,330;
// methods/DKT_regression.py:53-56,92).
#include "dkt_mll.h"
#include "dkt_tiles.h"

namespace {

struct Masks { bool row_ok, col_ok, is_acol; };

// Buffer addressing = one per-lane VGPR offset + a wave-uniform SGPR offset (no 64-bit per-element address
// registers for the compiler to hoist out of the class loop).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t brsrc;
__device__ __forceinline__ brsrc make_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float bload(brsrc r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(brsrc r, float v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

constexpr int ULD = 24;   // LDS row stride (floats) of a 16-column chunk of U: 16 + 8 -> conflict-free b128 fragments

template <int NT>
struct RegCtx {
    float* colbuf;   // [4][16*NT]
    int N, tx, ty, tid;
    bool col_ok;
};

// Element (pi, ji) of the register matrix.  Rows are held in PAIRS (pi = 2m, 2m + 1) so that the rank-1 update runs
// on v_pk_fma_f32: the row-factor pair comes straight out of one ds_read2_b32, the column factor is broadcast.
#define AE(pi, ji) A2[(pi) >> 1][ji][(pi) & 1]

// Pivot column as a thread sees it: its NT row entries (pairs), its column entries and the pivot itself.
template <int NT>
struct PivCol {
    f32x2 cp2[(NT + 1) / 2];
    float cj[NT];
    float d;
};

template <int NT, int KQ>
__device__ __forceinline__ void load_pivcol(PivCol<NT>& pc, const float* cb, int tx, int ty, int kr) {
    constexpr int NP2 = (NT + 1) / 2;
    pc.d = cb[16 * KQ + kr];
#pragma unroll
    for (int m = 0; m < NP2; ++m) {
        pc.cp2[m][0] = cb[ty + 32 * m];
        pc.cp2[m][1] = (2 * m + 1 < NT) ? cb[ty + 32 * m + 16] : 0.f;
    }
#pragma unroll
    for (int ji = KQ; ji < NT; ++ji) pc.cj[ji] = cb[tx + 16 * ji];
}

// One pivot step; PAR = kr & 1 is static so both LDS buffers have compile-time addresses.
// `cur` holds column k (read from LDS during the previous step); the step updates block column KQ, publishes column
// k+1, crosses the barrier, ISSUES the LDS reads of column k+1 into `nxt`, and only then updates the remaining block
// columns with column k -- the LDS round trip of the next step hides behind this step's bulk FMAs.
template <int NT, int KQ, int PAR>
__device__ __forceinline__ void sweep_step(f32x2 (&A2)[(NT + 1) / 2][NT], const RegCtx<NT>& c, PivCol<NT>& cur, PivCol<NT>& nxt,
                                           const int kr, const int kend, const bool lower_eq) {
    constexpr int NP = 16 * NT;
    constexpr int NP2 = (NT + 1) / 2;
    const int tx = c.tx, ty = c.ty;
    const float nrd = -__builtin_amdgcn_rcpf(cur.d);                  // -1 / d
    f32x2 cp2[NP2];
    float cj[NT];
    // row factor -(column / d); the column factor stays raw (product = L_pk L_jk)
#pragma unroll
    for (int m = 0; m < NP2; ++m) { cp2[m][0] = cur.cp2[m][0] * nrd; cp2[m][1] = cur.cp2[m][1] * nrd; }    // scalar on purpose, see sweep_pair
#pragma unroll
    for (int ji = KQ; ji < NT; ++ji) cj[ji] = cur.cj[ji];
    const float cpK = (ty == kr) ? nrd : cp2[KQ >> 1][KQ & 1];        // row k itself seeds U_kj = -L_jk / d
    cj[KQ] = (tx > kr) ? cj[KQ] : 0.f;                                // only columns j > k are updated
    cj[NT - 1] = c.col_ok ? cj[NT - 1] : 0.f;                         // padding columns j >= N
    const bool row_le_k = ty <= kr;
    const float cpk_le = row_le_k ? cpK : 0.f;                        // row block KQ, ji > KQ: p <= k only
    const float cpk_dd = (row_le_k || lower_eq) ? cpK : 0.f;          // block (KQ, KQ): p <= k or p >= j
    // update of block column ji: rows pi < KQ (U part) and pi >= ji (L part) plain, row block KQ masked,
    // KQ < pi < ji (k < p < j) untouched
    auto column = [&](const int ji) {                                 // ji is an unrolled constant
        const f32x2 cjv = {cj[ji], cj[ji]};
        const float sK = (ji == KQ) ? cpk_dd : cpk_le;
#pragma unroll
        for (int m = 0; m < NP2; ++m) {
            const int p0 = 2 * m, p1 = 2 * m + 1;
            // 0 none, 1 plain, 2 row block KQ
            const int k0 = (p0 < KQ) ? 1 : (p0 == KQ) ? 2 : (p0 < ji) ? 0 : 1;
            const int k1 = (p1 >= NT) ? 0 : (p1 < KQ) ? 1 : (p1 == KQ) ? 2 : (p1 < ji) ? 0 : 1;
            if (k0 == 1 && k1 == 1) {
                A2[m][ji] = __builtin_elementwise_fma(cp2[m], cjv, A2[m][ji]);
            } else if (k0 != 0 && k1 != 0) {
                const f32x2 v = {k0 == 2 ? sK : cp2[m][0], k1 == 2 ? sK : cp2[m][1]};
                A2[m][ji] = __builtin_elementwise_fma(v, cjv, A2[m][ji]);
            } else if (k0 != 0) {
                A2[m][ji][0] = __builtin_fmaf(k0 == 2 ? sK : cp2[m][0], cj[ji], A2[m][ji][0]);
            } else if (k1 != 0) {
                A2[m][ji][1] = __builtin_fmaf(k1 == 2 ? sK : cp2[m][1], cj[ji], A2[m][ji][1]);
            }
        }
    };
    column(KQ);
    if (kr + 1 < kend) {
        float* nb = c.colbuf + (PAR ^ 1) * NP;
        if (tx == kr + 1) {
#pragma unroll
            for (int pi = 0; pi < NT; ++pi) nb[ty + 16 * pi] = AE(pi, KQ);
        }
        __syncthreads();
        load_pivcol<NT, KQ>(nxt, nb, tx, ty, kr + 1);
    }
#pragma unroll
    for (int ji = KQ + 1; ji < NT; ++ji) column(ji);
}

// TWO pivots per barrier (k = 16 KQ + kr and k + 1, kr even).  Both raw columns are published together; every thread
// applies pivot k to its entries of column k+1 itself (x1' = x1 + F0 e with e = A[k+1][k]; the second pivot
// d1 = A[k+1][k+1] - e^2 / d0 is a uniform scalar), then does ONE rank-2 update.  The serial chain
// barrier -> LDS round trip -> rcp -> update of block column KQ -> publish is paid once per two pivots.
//   row factors   F0[p] = x0[p]  (seed 1 at p = k),     F1[p] = x1'[p] = x1[p] + F0[p] (-e/d0)  (seed 1 at p = k+1)
//   column factors y0s[j] = y0[j] (-1/d0) (j > k),   y1s[j] = (y1[j] + y0[j] (-e/d0)) (-1/d1)  (j > k+1)
//   A[p][j] += M0(p,j) F0[p] y0s[j] + M1(p,j) F1[p] y1s[j],   Mq(p,j) = (p <= k+q) or (p >= j)
template <int NT, int KQ, int PAR>
__device__ __forceinline__ void sweep_pair(f32x2 (&A2)[(NT + 1) / 2][NT], const RegCtx<NT>& c, const int kr, const int kend,
                                           const bool lower_eq) {
    constexpr int NP = 16 * NT;
    constexpr int NP2 = (NT + 1) / 2;
    const int tx = c.tx, ty = c.ty;
    const float* p0 = c.colbuf + (2 * PAR) * NP;          // column k
    const float* p1 = c.colbuf + (2 * PAR + 1) * NP;      // column k + 1
    __syncthreads();
    __builtin_amdgcn_s_setprio(3);                        // critical path (pivot columns -> next publish) outranks bulk updates
    const float d0 = p0[16 * KQ + kr], e = p0[16 * KQ + kr + 1], d1raw = p1[16 * KQ + kr + 1];
    f32x2 F0[NP2], F1[NP2];
    float y0[NT], y1[NT];
#pragma unroll
    for (int m = 0; m < NP2; ++m) {
        F0[m][0] = p0[ty + 32 * m];
        F0[m][1] = (2 * m + 1 < NT) ? p0[ty + 32 * m + 16] : 0.f;
        F1[m][0] = p1[ty + 32 * m];
        F1[m][1] = (2 * m + 1 < NT) ? p1[ty + 32 * m + 16] : 0.f;
    }
#pragma unroll
    for (int ji = KQ; ji < NT; ++ji) {
        y0[ji] = p0[tx + 16 * ji];
        y1[ji] = p1[tx + 16 * ji];
    }
    const float nrd0 = -__builtin_amdgcn_rcpf(d0);
    const float g0 = e * nrd0;                                         // -e / d0
    const float d1 = __builtin_fmaf(g0, e, d1raw);
    const float nrd1 = -__builtin_amdgcn_rcpf(d1);
    // The row factors stay RAW -- x0 and x1' = x1 + x0 (-e/d0) -- and the pivot reciprocals are folded into the COLUMN factors,
    // of which only the NT - KQ block columns from the pivot block on exist:  y0s = y0 (-1/d0),  y1s = (y1 + y0 (-e/d0)) (-1/d1).
    // (Scalar v_fma on purpose: with a split Gram kernel co-resident on the CU, the packed forms of the row-factor operations
    // were observed to round single 16-lane passes differently from run to run -- tools/corun_check.py)
    F0[KQ >> 1][KQ & 1] = (ty == kr) ? 1.f : F0[KQ >> 1][KQ & 1];       // row k seeds U_kj = -L_jk / d0
#pragma unroll
    for (int m = 0; m < NP2; ++m) {
        F1[m][0] = __builtin_fmaf(F0[m][0], g0, F1[m][0]);
        F1[m][1] = __builtin_fmaf(F0[m][1], g0, F1[m][1]);
    }
    F1[KQ >> 1][KQ & 1] = (ty == kr + 1) ? 1.f : F1[KQ >> 1][KQ & 1];   // row k+1 seeds U_{k+1,j}
#pragma unroll
    for (int ji = KQ; ji < NT; ++ji) {
        y1[ji] = __builtin_fmaf(y0[ji], g0, y1[ji]) * nrd1;
        y0[ji] *= nrd0;
    }
    y0[KQ] = (tx > kr) ? y0[KQ] : 0.f;                                 // pivot k updates columns j > k
    y1[KQ] = (tx > kr + 1) ? y1[KQ] : 0.f;                             // pivot k+1 updates columns j > k+1
    y0[NT - 1] = c.col_ok ? y0[NT - 1] : 0.f;                          // padding columns j >= N
    y1[NT - 1] = c.col_ok ? y1[NT - 1] : 0.f;
    const bool le0 = ty <= kr, le1 = ty <= kr + 1;
    const float f0K = F0[KQ >> 1][KQ & 1], f1K = F1[KQ >> 1][KQ & 1];
    const float f0_le = le0 ? f0K : 0.f, f0_dd = (le0 || lower_eq) ? f0K : 0.f;
    const float f1_le = le1 ? f1K : 0.f, f1_dd = (le1 || lower_eq) ? f1K : 0.f;
    auto column = [&](const int ji) {                                  // ji is an unrolled constant
        const f32x2 y0v = {y0[ji], y0[ji]}, y1v = {y1[ji], y1[ji]};
        const float s0 = (ji == KQ) ? f0_dd : f0_le, s1 = (ji == KQ) ? f1_dd : f1_le;
#pragma unroll
        for (int m = 0; m < NP2; ++m) {
            const int q0 = 2 * m, q1 = 2 * m + 1;
            // 0 none, 1 plain, 2 row block KQ
            const int k0 = (q0 < KQ) ? 1 : (q0 == KQ) ? 2 : (q0 < ji) ? 0 : 1;
            const int k1 = (q1 >= NT) ? 0 : (q1 < KQ) ? 1 : (q1 == KQ) ? 2 : (q1 < ji) ? 0 : 1;
            if (k0 == 1 && k1 == 1) {
                A2[m][ji] = __builtin_elementwise_fma(F0[m], y0v, A2[m][ji]);
                A2[m][ji] = __builtin_elementwise_fma(F1[m], y1v, A2[m][ji]);
            } else if (k0 != 0 && k1 != 0) {
                const f32x2 v0 = {k0 == 2 ? s0 : F0[m][0], k1 == 2 ? s0 : F0[m][1]};
                const f32x2 v1 = {k0 == 2 ? s1 : F1[m][0], k1 == 2 ? s1 : F1[m][1]};
                A2[m][ji] = __builtin_elementwise_fma(v0, y0v, A2[m][ji]);
                A2[m][ji] = __builtin_elementwise_fma(v1, y1v, A2[m][ji]);
            } else if (k0 != 0) {
                A2[m][ji][0] = __builtin_fmaf(k0 == 2 ? s0 : F0[m][0], y0[ji], A2[m][ji][0]);
                A2[m][ji][0] = __builtin_fmaf(k0 == 2 ? s1 : F1[m][0], y1[ji], A2[m][ji][0]);
            } else if (k1 != 0) {
                A2[m][ji][1] = __builtin_fmaf(k1 == 2 ? s0 : F0[m][1], y0[ji], A2[m][ji][1]);
                A2[m][ji][1] = __builtin_fmaf(k1 == 2 ? s1 : F1[m][1], y1[ji], A2[m][ji][1]);
            }
        }
    };
    column(KQ);
    if (kr + 3 < kend) {                                               // another full pair follows: publish it now
        float* n0 = c.colbuf + (2 * (PAR ^ 1)) * NP;
        if (tx == kr + 2 || tx == kr + 3) {
            float* nb = n0 + (tx - (kr + 2)) * NP;
#pragma unroll
            for (int pi = 0; pi < NT; ++pi) nb[ty + 16 * pi] = AE(pi, KQ);
        }
    }
    __builtin_amdgcn_s_setprio(1);                        // bulk updates of a sweep still outrank the gradient product phases
#pragma unroll
    for (int ji = KQ + 1; ji < NT; ++ji) column(ji);
}

// One block column KQ of the sweep: k = 16*KQ + kr, kr = 0 .. min(16, N - 16*KQ) - 1.
// Columns are scaled LAZILY (the registers keep the raw columns; 1 / L_kk is applied once after the sweep), and the
// pivots themselves are not inspected here: d_k stays in the diagonal slot (k, k), from which the caller takes
// log det, the column scales and the first non-positive pivot after the sweep.
template <int NT, int KQ>
__device__ __forceinline__ void sweep_block(f32x2 (&A2)[(NT + 1) / 2][NT], const RegCtx<NT>& c) {
    constexpr int NP = 16 * NT;
    const int kend = min(16, c.N - 16 * KQ);
    const int tx = c.tx, ty = c.ty;
    const bool lower_eq = ty >= tx;
    // Diagonal blocks below the pivot block are updated in BOTH triangles (no per-step mask); the strictly-upper
    // half then holds Schur-complement values nobody reads, and is cleared here, when the block becomes the pivot
    // block and its upper half starts to collect U.
    if constexpr (KQ > 0) AE(KQ, KQ) = lower_eq ? AE(KQ, KQ) : 0.f;
#if !defined(DKT_MLL_SINGLE_STEP)
    const int npair2 = kend & ~1;                        // pivots handled two at a time
    if (npair2 > 0) {
        if (tx < 2) {
            float* nb = c.colbuf + tx * NP;
#pragma unroll
            for (int pi = 0; pi < NT; ++pi) nb[ty + 16 * pi] = AE(pi, KQ);
        }
        for (int kr = 0; kr < npair2; kr += 4) {
            sweep_pair<NT, KQ, 0>(A2, c, kr, npair2, lower_eq);
            if (kr + 2 < npair2) sweep_pair<NT, KQ, 1>(A2, c, kr + 2, npair2, lower_eq);
        }
    }
    if (kend & 1) {                                      // odd leftover pivot of the last block
        const int kr = kend - 1;
        // a vector the last pair did NOT read: pair p reads vectors 2 (p & 1), 2 (p & 1) + 1; the last pair is p = npair2/2 - 1
        float* sb = c.colbuf + (((npair2 >> 1) & 1) ? 2 : 0) * NP;
        if (tx == kr) {
#pragma unroll
            for (int pi = 0; pi < NT; ++pi) sb[ty + 16 * pi] = AE(pi, KQ);
        }
        __syncthreads();
        PivCol<NT> s0, s1;
        load_pivcol<NT, KQ>(s0, sb, tx, ty, kr);
        sweep_step<NT, KQ, 0>(A2, c, s0, s1, kr, kend, lower_eq);
    }
#else
    if (tx == 0) {
#pragma unroll
        for (int pi = 0; pi < NT; ++pi) c.colbuf[ty + 16 * pi] = AE(pi, KQ);
    }
    __syncthreads();
    PivCol<NT> s0, s1;
    load_pivcol<NT, KQ>(s0, c.colbuf, tx, ty, 0);
    for (int kr = 0; kr < kend; kr += 2) {
        sweep_step<NT, KQ, 0>(A2, c, s0, s1, kr, kend, lower_eq);
        if (kr + 1 < kend) sweep_step<NT, KQ, 1>(A2, c, s1, s0, kr + 1, kend, lower_eq);
    }
#endif
}

template <int NT, int KQ>
__device__ __forceinline__ void sweep_all(f32x2 (&A2)[(NT + 1) / 2][NT], const RegCtx<NT>& c) {
    if constexpr (KQ < NT) {
        if (16 * KQ >= c.N) return;
        sweep_block<NT, KQ>(A2, c);
        sweep_all<NT, KQ + 1>(A2, c);
    }
}

// Five block-wide sums at once (2 barriers).  red: >= 20 floats.
__device__ __forceinline__ void block_sum5(float (&v)[5], float* red) {
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) red[(threadIdx.x >> 6) * 5 + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = red[i] + red[5 + i] + red[10 + i] + red[15 + i];
}

// MFMA accumulation of one 16-column chunk of U U^T for the tile rows RA (and RB >= 0) owned by this wave.
// acc index: tiles of the SHORT row RB first (tj = 0..RB), then the tiles of row RA (tj = 0..RA).  Row block R of U is
// zero left of column chunk R, so its tiles are first touched -- with a zero C operand -- at chunk R: no accumulator
// is live before it is needed (the long rows start late, when most of the matrix registers are already dead).
template <int NT, int RA, int RB, int CH>
__device__ __forceinline__ void w_chunk_mfma(f32x4* acc, const float* ub, int r16, int q) {
    constexpr int OA = (RB >= 0) ? RB + 1 : 0;
    const float* base = ub + r16 * ULD + 4 * q;
    if constexpr (RB >= 0 && RB <= CH) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + RB * 16 * ULD);
#pragma unroll
        for (int tj = 0; tj <= RB; ++tj) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(base + tj * 16 * ULD);
            f32x4 cacc = (CH == RB) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[tj];
#pragma unroll
            for (int t = 0; t < 4; ++t) cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bf[t], cacc, 0, 0, 0);
            acc[tj] = cacc;
        }
    }
    if constexpr (RA <= CH) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(base + RA * 16 * ULD);
#pragma unroll
        for (int tj = 0; tj <= RA; ++tj) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(base + tj * 16 * ULD);
            f32x4 cacc = (CH == RA) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[OA + tj];
#pragma unroll
            for (int t = 0; t < 4; ++t) cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bf[t], cacc, 0, 0, 0);
            acc[OA + tj] = cacc;
        }
    }
}

// W tile = coef (alpha_i alpha_j - (U U^T)_ij), added into W (lower triangle + mirror); the first class stores
template <int NT, int ROW>
__device__ __forceinline__ void w_accum_row(const f32x4* acc, brsrc Wr, int N, int tyN, int r16, int q, int vo_rc, int vo_cr,
                                            bool first, bool last, float coef, const float* alv) {
    const f32x4 ai = *reinterpret_cast<const f32x4*>(alv + 16 * ROW + 4 * q) * coef;
    // read-modify-write of the running sum over the classes: ALL loads of the tile row are issued before the first store
    // (the compiler may not move a load above a store to the same buffer, so the obvious per-element form pays one
    // memory round trip per element: 28 serialized latencies per wave and class)
    float prev[ROW + 1][4];
#pragma unroll
    for (int tj = 0; tj <= ROW; ++tj) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int pl = 4 * q + reg;
            bool ok = true;
            if (ROW == NT - 1) ok = ok && (pl < tyN);
            if (tj == NT - 1) ok = ok && (r16 < tyN);
            if (tj == ROW) ok = ok && (r16 <= pl);
            // an out-of-range offset reads as 0 through the descriptor (masked lanes, first class)
            prev[tj][reg] = bload(Wr, (ok && !first) ? vo_rc : 0x7ffffff0, ((16 * ROW + reg) * N + 16 * tj) * 4);
        }
    }
#pragma unroll
    for (int tj = 0; tj <= ROW; ++tj) {
        const float aj = alv[16 * tj + r16];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int pl = 4 * q + reg;
            bool ok = true;
            if (ROW == NT - 1) ok = ok && (pl < tyN);                    // gi < N
            if (tj == NT - 1) ok = ok && (r16 < tyN);                    // gj < N
            if (tj == ROW) ok = ok && (r16 <= pl);                       // lower triangle of the diagonal tile
            if (ok) {
                const int so = ((16 * ROW + reg) * N + 16 * tj) * 4;
                const float v = __builtin_fmaf(ai[reg], aj, -coef * acc[tj][reg]) + prev[tj][reg];
                bstore(Wr, v, vo_rc, so);
                // the strided mirror write happens once, when the sum over the classes is complete
                if (last && !(tj == ROW && r16 == pl)) bstore(Wr, v, vo_cr, (16 * tj * N + 16 * ROW + reg) * 4);
            }
        }
    }
}

// All chunks of the product for one class, for the wave that owns tile rows RA / RB: every thread writes its share of
// the 16-column chunk CH of U to LDS (one barrier per chunk, double buffered), then the wave accumulates its tile rows.
// The four waves run four different instantiations (the caller switches on the wave id); each executes exactly NT
// barriers, so the workgroup barrier still pairs up chunk by chunk.
template <int NT, int RA, int RB, int CH>
__device__ __forceinline__ void w_product_all(f32x4* acc, f32x2 (&A2)[(NT + 1) / 2][NT], float* ubuf, int tx, int ty,
                                              int r16, int q, const Masks& m) {
    if constexpr (CH < NT) {
        constexpr int NP = 16 * NT;
        float* ub = ubuf + (CH & 1) * NP * ULD;
#pragma unroll
        for (int pi = 0; pi < NT; ++pi) {
            const int p = ty + 16 * pi;
            float v = 0.f;
            if (CH > pi) v = AE(pi, CH);                               // U block
            else if (CH == pi) v = (tx >= ty) ? AE(pi, CH) : 0.f;      // diagonal block: upper incl. diagonal
            if (CH == NT - 1) v = m.col_ok ? v : 0.f;                  // columns >= N are not U
            if (pi == NT - 1) v = m.row_ok ? v : 0.f;                  // rows >= N
            ub[p * ULD + tx] = v;
        }
        __syncthreads();
        if constexpr (RA >= 0) w_chunk_mfma<NT, RA, RB, CH>(acc, ub, r16, q);
        w_product_all<NT, RA, RB, CH + 1>(acc, A2, ubuf, tx, ty, r16, q, m);
    }
}

template <int NT, int W>
__device__ __forceinline__ void w_product_wave(f32x2 (&A2)[(NT + 1) / 2][NT], float* ubuf, int tx, int ty, int r16, int q, const Masks& m,
                                               brsrc Wr, int N, int tyN, int vo_rc, int vo_cr, bool first, bool last, float coef, const float* alv) {
    constexpr int RA = RowsOf<NT, W>::RA, RB = RowsOf<NT, W>::RB;
    f32x4 acc[NT + 1];                                   // first touched (zero C operand) inside the product
    w_product_all<NT, RA, RB, 0>(acc, A2, ubuf, tx, ty, r16, q, m);
    if constexpr (RB >= 0) w_accum_row<NT, RB>(acc, Wr, N, tyN, r16, q, vo_rc, vo_cr, first, last, coef, alv);
    if constexpr (RA >= 0) w_accum_row<NT, RA>(acc + (RB >= 0 ? RB + 1 : 0), Wr, N, tyN, r16, q, vo_rc, vo_cr, first, last, coef, alv);
}

template <int NT, bool WANT_GRAD, bool WANT_CHOL>
#ifndef DKT_REG_MINW
#define DKT_REG_MINW 4
#endif
__global__ __launch_bounds__(256, ((NT <= 7 && !(WANT_GRAD && WANT_CHOL)) ? DKT_REG_MINW : 2)) void mll_reg_kernel(MllArgs a) {
    constexpr int NP = 16 * NT;
    __shared__ float colbuf[4 * NP];           // two double-buffered pivot-column pairs
    __shared__ float wv[NP];
    __shared__ float dgv[NP];
    __shared__ __attribute__((aligned(16))) float alv[NP];
    __shared__ float red[20];
    __shared__ __attribute__((aligned(16))) float ubuf[2 * NP * ULD];

    const int b = blockIdx.x, tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int lane = tid & 63, r16 = lane & 15, q = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave id in an SGPR: scalar switch
    const int N = a.N, C = a.C;
    const int tyN = N - 16 * (NT - 1);                  // row N lives at pi = NT-1, ty = tyN; column N at ji = NT-1, tx = tyN
    const bool lower_eq = ty >= tx, upper_eq = tx >= ty;
    const bool row_ok = ty < tyN, is_w = ty == tyN;      // last block row: p < N / p == N
    const bool col_ok = tx < tyN, is_acol = tx == tyN;    // last block column: j < N / j == N
    const brsrc Er = make_rsrc(a.E + (size_t)b * N * N, N * N * 4);
    const int vo_form = (ty * N + tx) * 4;               // per-lane part of element (ty + 16 pi, tx + 16 ji)
    const int vo_rc = (4 * q * N + r16) * 4, vo_cr = (r16 * N + 4 * q) * 4;   // MFMA-layout (row 4q, col r16) and its mirror
    constexpr bool want_grad = WANT_GRAD;
    constexpr bool want_chol = WANT_CHOL;

    Masks masks;
    masks.row_ok = row_ok; masks.col_ok = col_ok; masks.is_acol = is_acol;
    RegCtx<NT> ctx;
    ctx.colbuf = colbuf; ctx.N = N; ctx.tx = tx; ctx.ty = ty; ctx.tid = tid; ctx.col_ok = col_ok;

    bool poisoned = false;

    for (int c = 0; c < C; ++c) {
        const float svc = a.sv[c], mc = a.mean[c], nzc = a.noise[c];
        const brsrc yr = make_rsrc(a.Y + (size_t)b * a.y_bstride + (size_t)c * N, N * 4);
        f32x2 A2[(NT + 1) / 2][NT];
        float rinvcol[NT];
        float log2part = 0.f;                            // this thread's share of sum_k log2 d_k (ty == 0 lanes)
        int fail_at = 0;
        float jit = 0.f;
        for (int attempt = 0; attempt <= a.max_tries; ++attempt) {
            jit = 0.f;
            if (attempt > 0) {
                jit = a.jitter0;
                for (int i = 1; i < attempt; ++i) jit *= 10.f;
            }
#pragma unroll
            for (int pi = 0; pi < NT; ++pi) {
#pragma unroll
                for (int ji = 0; ji < NT; ++ji) {
                    // static block structure; only the last block row / column need per-thread masks
                    float v = 0.f;
                    if (pi >= ji) {
                        bool ld = true;                                    // lower-triangle element of K?
                        if (pi == ji) ld = lower_eq;
                        if (pi == NT - 1) ld = ld && row_ok;
                        if (ji == NT - 1) ld = ld && col_ok;
                        {
                            float x = svc * bload(Er, vo_form, (16 * pi * N + 16 * ji) * 4);   // out-of-range reads return 0
                            if (pi == ji && tx == ty) x += nzc + jit;
                            v = ld ? x : 0.f;
                        }
                        if (pi == NT - 1) {
                            bool lw = is_w;
                            if (ji == NT - 1) lw = lw && col_ok;
                            const float yv = bload(yr, tx * 4, 16 * ji * 4) - mc;
                            v = lw ? yv : v;
                        }
                    }
                    AE(pi, ji) = v;
                }
            }
            __syncthreads();          // previous users of colbuf are done
            sweep_all<NT, 0>(A2, ctx);
            __builtin_amdgcn_s_setprio(0);
            // ---- pivots: d_j sits raw in the diagonal slot (j, j).  Column scales, log det, first bad pivot ----
            if (ty == tx) {
#pragma unroll
                for (int ji = 0; ji < NT; ++ji) dgv[tx + 16 * ji] = AE(ji, ji);
            }
            __syncthreads();
            int bad = 0x7fffffff;
            log2part = 0.f;
#pragma unroll
            for (int ji = NT - 1; ji >= 0; --ji) {
                const float dj = dgv[tx + 16 * ji];
                const bool valid = (ji < NT - 1) || col_ok;
                rinvcol[ji] = valid ? __builtin_amdgcn_rsqf(dj) : 1.0f;
                log2part += valid ? __builtin_amdgcn_logf(dj) : 0.f;          // v_log_f32 = log2
                bad = (valid && !(dj > 0.f)) ? tx + 16 * ji + 1 : bad;        // descending ji: the smallest index wins
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) bad = min(bad, __shfl_xor(bad, o, DKT_WAVE));
            bad = __builtin_amdgcn_readfirstlane(bad);       // every 16-lane row of every wave computed the same value
            fail_at = (bad == 0x7fffffff) ? 0 : bad;
            if (fail_at == 0) break;
        }
        const size_t bc = (size_t)b * C + c;
        if (fail_at != 0) {
            const float qnan = __int_as_float(0x7fc00000);
            if (tid == 0) {
                a.logp[bc] = qnan;
                a.jitter_used[bc] = jit;
                a.info[bc] = fail_at;
                if (want_grad) { a.dsv[bc] = qnan; a.dmean[bc] = qnan; a.dnoise[bc] = qnan; }
            }
            for (int i = tid; i < N; i += 256) a.alpha[bc * N + i] = qnan;
            if constexpr (want_chol)
                for (int idx = tid; idx < N * N; idx += 256) a.L[bc * N * N + idx] = qnan;
            poisoned = true;
            __syncthreads();
            continue;
        }
        // ---- apply the lazy column scaling: column k gets 1 / L_kk, the diagonal slot becomes U_kk = 1 / L_kk ----
#pragma unroll
        for (int pi = 0; pi < NT; ++pi) {
#pragma unroll
            for (int ji = 0; ji < NT; ++ji) {
                AE(pi, ji) *= rinvcol[ji];
                if (pi == ji) {
                    bool dg = tx == ty;
                    if (pi == NT - 1) dg = dg && col_ok;           // the (N, N) corner and beyond are padding, not pivots
                    AE(pi, ji) = dg ? rinvcol[ji] : AE(pi, ji);
                }
            }
        }
        // ---- w row -> LDS; scalars ----
        if (ty == tyN) {
#pragma unroll
            for (int ji = 0; ji < NT; ++ji) wv[tx + 16 * ji] = AE(NT - 1, ji);
        }
        __syncthreads();
        float wj[NT];
#pragma unroll
        for (int ji = 0; ji < NT; ++ji) wj[ji] = wv[tx + 16 * ji];
        float v5[5] = {0.f, (ty == 0) ? log2part : 0.f, 0.f, 0.f, 0.f};     // quad, sum log2 d, sum alpha, sum alpha^2, |U|_F^2
        if (ty == tyN) {
#pragma unroll
            for (int ji = 0; ji < NT; ++ji) v5[0] += wj[ji] * wj[ji];
        }
#pragma unroll
        for (int pi = 0; pi < NT; ++pi) {
            const int p = ty + 16 * pi;
            float s = 0.f, u2 = 0.f;
#pragma unroll
            for (int ji = pi; ji < NT; ++ji) {
                float u = AE(pi, ji);
                if (ji == pi) u = upper_eq ? u : 0.f;              // strictly-lower entries of the diagonal block are L
                if (ji == NT - 1) u = col_ok ? u : 0.f;            // padding / alpha slot
                s += u * wj[ji];
                u2 += u * u;
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, DKT_WAVE);
            if (pi == NT - 1) { s = row_ok ? s : 0.f; u2 = row_ok ? u2 : 0.f; }
            v5[4] += u2;
            if (tx == 0) {
                alv[p] = s;                                         // alpha_p for the gradient product (column N of [U | alpha])
                v5[2] += s;
                v5[3] += s * s;
                if (pi < NT - 1 || row_ok) a.alpha[bc * N + p] = s;
            }
        }
        block_sum5(v5, red);
        const float quad = v5[0], logdet_half = 0.34657359027997264f * v5[1], asum = v5[2], a2 = v5[3], trk = v5[4];
        if (tid == 0) {
            a.logp[bc] = -0.5f * quad - logdet_half - (float)N * DKT_HALF_LOG_2PI;
            a.jitter_used[bc] = jit;
            a.info[bc] = 0;
            if (want_grad) {
                const float nz_eff = nzc + jit;
                a.dmean[bc] = asum;
                a.dnoise[bc] = 0.5f * (a2 - trk);
                a.dsv[bc] = 0.5f * ((quad - (float)N) - nz_eff * (a2 - trk)) / svc;
            }
        }
        if constexpr (want_chol) {
            float* Lb = a.L + bc * N * N;
#pragma unroll
            for (int pi = 0; pi < NT; ++pi) {
#pragma unroll
                for (int ji = 0; ji < NT; ++ji) {
                    const int p = ty + 16 * pi, j = tx + 16 * ji;
                    bool ok = true;
                    if (pi == NT - 1) ok = ok && row_ok;
                    if (ji == NT - 1) ok = ok && col_ok;
                    if (ok) {
                        float v = 0.f;
                        if (pi > ji) v = AE(pi, ji);
                        else if (pi == ji) v = (ty > tx) ? AE(pi, ji) : ((ty == tx) ? 1.0f / rinvcol[ji] : 0.f);
                        Lb[p * N + j] = v;
                    }
                }
            }
        }
        if constexpr (want_grad) {
            const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
            const float coef = 0.5f * cw * svc;
            // W += coef (alpha alpha^T - U U^T): U U^T on MFMA, the rank-1 alpha term in the epilogue (alpha from LDS)
            // The class's contribution is accumulated into W[b] in memory (same lanes, same words, class after class:
            // deterministic, no atomics) so that no accumulator registers stay live across the next class's sweep.
            const brsrc Wr = make_rsrc(a.W + (size_t)b * N * N, N * N * 4);
            const bool first = (c == 0), last = (c == C - 1);
            switch (wave_u) {
                case 0: w_product_wave<NT, 0>(A2, ubuf, tx, ty, r16, q, masks, Wr, N, tyN, vo_rc, vo_cr, first, last, coef, alv); break;
                case 1: w_product_wave<NT, 1>(A2, ubuf, tx, ty, r16, q, masks, Wr, N, tyN, vo_rc, vo_cr, first, last, coef, alv); break;
                case 2: w_product_wave<NT, 2>(A2, ubuf, tx, ty, r16, q, masks, Wr, N, tyN, vo_rc, vo_cr, first, last, coef, alv); break;
                default: w_product_wave<NT, 3>(A2, ubuf, tx, ty, r16, q, masks, Wr, N, tyN, vo_rc, vo_cr, first, last, coef, alv); break;
            }
        }
        __syncthreads();
    }

    if constexpr (want_grad) {
        if (poisoned) {
            float* Wb = a.W + (size_t)b * N * N;
            const float qnan = __int_as_float(0x7fc00000);
            for (int idx = tid; idx < N * N; idx += 256) Wb[idx] = qnan;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// chol_inv_block_kernel<NT>: the same sweep on a batch of nb x nb SPD blocks that live inside larger matrices (leading
// dimension ld): block A (lower triangle read) -> L (lower incl. diagonal) and U = L^-T (upper incl. diagonal = 1 / L_kk).
// Building block of the blocked path for N > 127 (dkt_mll_big.hip): with U at hand the panel step L_ij = A_ij U_jj is a GEMM.
// info[m] receives pivot_base + (index of the first non-positive pivot) + 1 unless an earlier block already failed.
template <int NT>
__global__ __launch_bounds__(256, NT <= 7 ? 4 : 2) void chol_inv_block_kernel(const float* __restrict__ A, int lda, long sA,
                                                                             float* __restrict__ L, int ldl, long sL,
                                                                             float* __restrict__ U, int ldu, long sU, int nb,
                                                                             int pivot_base, int32_t* __restrict__ info) {
    constexpr int NP = 16 * NT;
    __shared__ float colbuf[4 * NP];
    __shared__ float dgv[NP];
    const int m = blockIdx.x, tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const float* Ab = A + (size_t)m * sA;
    float* Lb = L + (size_t)m * sL;
    float* Ub = U + (size_t)m * sU;
    const int tyN = nb - 16 * (NT - 1);                  // valid columns of the last block column (row nb is an unused zero row)
    const bool col_ok = tx < tyN;
    f32x2 A2[(NT + 1) / 2][NT];
#pragma unroll
    for (int pi = 0; pi < NT; ++pi) {
#pragma unroll
        for (int ji = 0; ji < NT; ++ji) {
            const int p = ty + 16 * pi, j = tx + 16 * ji;
            float v = 0.f;
            if (pi >= ji && p >= j && p < nb && j < nb) v = Ab[(size_t)p * lda + j];
            AE(pi, ji) = v;
        }
    }
    RegCtx<NT> ctx;
    ctx.colbuf = colbuf; ctx.N = nb; ctx.tx = tx; ctx.ty = ty; ctx.tid = tid; ctx.col_ok = col_ok;
    __syncthreads();
    sweep_all<NT, 0>(A2, ctx);
    __builtin_amdgcn_s_setprio(0);
    if (ty == tx) {
#pragma unroll
        for (int ji = 0; ji < NT; ++ji) dgv[tx + 16 * ji] = AE(ji, ji);
    }
    __syncthreads();
    float rinvcol[NT];
    int bad = 0x7fffffff;
#pragma unroll
    for (int ji = NT - 1; ji >= 0; --ji) {
        const float dj = dgv[tx + 16 * ji];
        const bool valid = (ji < NT - 1) || col_ok;
        rinvcol[ji] = valid ? __builtin_amdgcn_rsqf(dj) : 1.0f;
        bad = (valid && !(dj > 0.f)) ? tx + 16 * ji + 1 : bad;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) bad = min(bad, __shfl_xor(bad, o, DKT_WAVE));
    if (tid == 0 && bad != 0x7fffffff && info[m] == 0) info[m] = pivot_base + bad;
#pragma unroll
    for (int pi = 0; pi < NT; ++pi) {
#pragma unroll
        for (int ji = 0; ji < NT; ++ji) {
            const int p = ty + 16 * pi, j = tx + 16 * ji;
            if (p < nb && j < nb) {
                const float v = AE(pi, ji) * rinvcol[ji];
                if (p > j) {
                    Lb[(size_t)p * ldl + j] = v;
                    Ub[(size_t)p * ldu + j] = 0.f;          // consumers multiply with the full U block: its lower part must be 0
                } else if (p < j) Ub[(size_t)p * ldu + j] = v;
                else {
                    Lb[(size_t)p * ldl + j] = 1.0f / rinvcol[ji];
                    Ub[(size_t)p * ldu + j] = rinvcol[ji];
                }
            }
        }
    }
}

template <int NT>
void launch_reg(const MllArgs& a, hipStream_t st) {
    const bool g = (a.flags & DKT_MLL_WANT_GRAD) != 0, c = (a.flags & DKT_MLL_WANT_CHOL) != 0;
    if (g && c) hipLaunchKernelGGL((mll_reg_kernel<NT, true, true>), dim3(a.B), dim3(256), 0, st, a);
    else if (g) hipLaunchKernelGGL((mll_reg_kernel<NT, true, false>), dim3(a.B), dim3(256), 0, st, a);
    else if (c) hipLaunchKernelGGL((mll_reg_kernel<NT, false, true>), dim3(a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((mll_reg_kernel<NT, false, false>), dim3(a.B), dim3(256), 0, st, a);
}

}  // namespace

bool dkt_mll_reg_launch(const MllArgs& a, hipStream_t st) {
    const int nt = (a.N + 1 + 15) / 16;
    switch (nt) {
        case 1: launch_reg<1>(a, st); return true;
        case 2: launch_reg<2>(a, st); return true;
        case 3: launch_reg<3>(a, st); return true;
        case 4: launch_reg<4>(a, st); return true;
        case 5: launch_reg<5>(a, st); return true;
        case 6: launch_reg<6>(a, st); return true;
        case 7: launch_reg<7>(a, st); return true;
        case 8: launch_reg<8>(a, st); return true;
        default: return false;
    }
}

void dkt_chol_inv_block_launch(const float* A, int lda, long sA, float* L, int ldl, long sL, float* U, int ldu, long sU, int nb,
                               int pivot_base, int32_t* info, int nmat, hipStream_t st) {
    switch ((nb + 1 + 15) / 16) {
        case 1: hipLaunchKernelGGL((chol_inv_block_kernel<1>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 2: hipLaunchKernelGGL((chol_inv_block_kernel<2>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 3: hipLaunchKernelGGL((chol_inv_block_kernel<3>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 4: hipLaunchKernelGGL((chol_inv_block_kernel<4>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 5: hipLaunchKernelGGL((chol_inv_block_kernel<5>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 6: hipLaunchKernelGGL((chol_inv_block_kernel<6>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 7: hipLaunchKernelGGL((chol_inv_block_kernel<7>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        default: hipLaunchKernelGGL((chol_inv_block_kernel<8>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
    }
}
