// dkt_reg_sweep.h -- the register-resident sweep machinery (2-D cyclic working matrix over a 16 x 16 thread grid, one barrier and one LDS round trip per
// pivot) shared by chol_inv_block_kernel (dkt_mll_reg.hip: diagonal blocks of the blocked path for N > 127) and by the round-1 marginal-likelihood kernel
// mll_reg_kernel, which since round 4 lives in the measurement library only (dkt_mll_reg_twin.hip -> libdkt_diag.so).  Description: dkt_mll_reg_twin.hip.
#pragma once
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

}  // namespace
