// dkt_mfma_tiles.h -- 16 x 16 tile primitives on v_mfma_f32_16x16x4_f32 shared by the marginal-likelihood kernels
// (dkt_mll_mfma.hip: wave-per-matrix, tiles in VGPRs; dkt_mll_tiled.hip: large N, tiles in a global tile array):
// buffer-descriptor loads / stores, D += X^T Y on accumulator-layout tiles, and the diagonal-tile sweep
// (replicated column layout, one DPP-fused FMA per row and pivot).  See the header of dkt_mll_mfma.hip for the algorithm.
#pragma once
#include "dkt_mll.h"

namespace dkt_mfma {

typedef __amdgpu_buffer_rsrc_t brsrc;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ brsrc mk_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(brsrc r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
// 16-byte stores: the scalar offset is folded into the VGPR offset and the instruction's soffset stays the literal 0.  With a REGISTER in the soffset
// field hipcc assumes that the store's data registers may be rewritten by the very next VALU instruction (its hazard recogniser skips that form) -- on
// gfx950 they may not: a loop that stored dX with soffset = 4 d0 wrote garbage that changed from run to run until the offset moved into the VGPR
// (round 5, the dropped software-pipelined fused backward; profiles/r05/v5_stage_ab.log).
__device__ __forceinline__ void bstore4(brsrc r, f32x4 x, int voff, int soff) {
    const u32x4 v = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)((unsigned)voff + (unsigned)soff), 0, 0);
}
__device__ __forceinline__ void bstore1(brsrc r, float x, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, 0);
}
// the same with the non-temporal hint (cache policy bit 1 = nt): a stream that is touched once per pass should not push the pass' reused operands out of L2
template <int AUX>
__device__ __forceinline__ f32x4 bload4_pol(brsrc r, int voff, int soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    return (f32x4){__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}
template <int AUX>
__device__ __forceinline__ void bstore4_pol(brsrc r, f32x4 x, int voff, int soff) {
    const u32x4 v = {__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3])};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)((unsigned)voff + (unsigned)soff), 0, AUX);
}
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
// x[i] += row_newbcast_P(x[i]) * t for rows i = I0 .. I0 + CNT - 1 as ONE asm block of v_fmac_f32_dpp (hipcc does not fuse the DPP
// move into the FMA: it emits v_mov 0 / s_nop / v_mov_dpp / v_fmac per update).  The leading s_nop 1 covers the "VALU write -> DPP
// read: 2 wait states" hazard for whatever the compiler placed just before the block; inside it every instruction reads a register
// written at least one pivot earlier.
#define DKT_FMD(k) "v_fmac_f32_dpp %" #k ", %" #k ", %[t] row_newbcast:%[p] row_mask:0xf bank_mask:0xf\n\t"
#ifdef DKT_SWEEP_NO_NOP          // experiment: no wait states in front of a piece (every x[i] it reads was written a pivot earlier)
#define DKT_PIECE_NOP ""
#else
#define DKT_PIECE_NOP "s_nop 1\n\t"
#endif
template <int P, int I0, int CNT>
__device__ __forceinline__ void sweep_rows_piece(float (&x)[16], const float t) {
    static_assert(CNT >= 0 && CNT <= 5 && I0 + CNT <= 16, "piece");
    if constexpr (CNT == 1)
        asm volatile(DKT_PIECE_NOP DKT_FMD(0) : "+v"(x[I0 + 0]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (CNT == 2)
        asm volatile(DKT_PIECE_NOP DKT_FMD(0) DKT_FMD(1) : "+v"(x[I0 + 0]), "+v"(x[I0 + 1]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (CNT == 3)
        asm volatile(DKT_PIECE_NOP DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) : "+v"(x[I0 + 0]), "+v"(x[I0 + 1]), "+v"(x[I0 + 2]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (CNT == 4)
        asm volatile(DKT_PIECE_NOP DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) : "+v"(x[I0 + 0]), "+v"(x[I0 + 1]), "+v"(x[I0 + 2]), "+v"(x[I0 + 3]) : [t] "v"(t), [p] "n"(P));
    else if constexpr (CNT == 5)
        asm volatile(DKT_PIECE_NOP DKT_FMD(0) DKT_FMD(1) DKT_FMD(2) DKT_FMD(3) DKT_FMD(4) : "+v"(x[I0 + 0]), "+v"(x[I0 + 1]), "+v"(x[I0 + 2]), "+v"(x[I0 + 3]), "+v"(x[I0 + 4]) : [t] "v"(t), [p] "n"(P));
}
#undef DKT_FMD

template <int P>
__device__ __forceinline__ float pivot_bcast(const float xp) {
    float d;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(xp), "n"(P));
    return d;
}

// One pivot: x holds -A.  t = x[P] / d is the update factor of the row (Schur lanes c > P and L^-1 lanes c < P alike); at lane P itself
// the multiplier column turns into a column of L^-1, x[i] <- x[i] / d = x[i] (1 + t) with t = (1 - d) / d -- no cancellation because
// the caller scales the matrix by a power of 4 so that every pivot is <= 1.
// No guard on d: a non-positive pivot turns the matrix into inf / NaN -- it is reported through dv and the caller poisons the
// outputs anyway -- except the augmented pivot of the last tile (LAST: local index pn, forced to 1; padding pivots are 1 already).
template <int P, bool LAST>
__device__ __forceinline__ float sweep_pivot_head(float (&x)[16], float& dv, const Lane& ln, const int pn) {
    float d = -pivot_bcast<P>(x[P]);                      // d_P, uniform
    const bool eq = ln.c == P;
    dv = eq ? d : dv;
    if constexpr (LAST) d = (P == pn) ? 1.0f : d;
    const float rs = __builtin_amdgcn_rsqf(d);            // 1 / sqrt(d)
    const float rs2 = rs * rs;
    const float t = (eq ? 1.0f - d : x[P]) * rs2;
    x[P] = eq ? rs : x[P] * rs;
    return t;
}

// accumulator layout -> replicated column layout
__device__ __forceinline__ void sweep_begin(const f32x4 S, float (&x)[16], float& dv) {
#pragma unroll
    for (int q = 0; q < 4; ++q) spread_rows(S[q], x[q], x[4 + q], x[8 + q], x[12 + q]);
    dv = 1.0f;
}

// swept rows -> M = R^-T in the accumulator layout
__device__ __forceinline__ f32x4 sweep_end(const float (&x)[16], const Lane& ln) {
    f32x4 M;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // (opaque copies: hipcc otherwise turns the select chain into an indexed load of x[4 g + q] from a scratch copy of x)
        float r0 = x[q], r1 = x[4 + q], r2 = x[8 + q], r3 = x[12 + q];
        asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        const float v = ln.g0 ? r0 : (ln.g1 ? r1 : (ln.g2 ? r2 : r3));
        M[q] = (ln.c <= 4 * ln.g + q) ? v : 0.f;
    }
    return M;
}

// Wave-wide sum / maximum without lane-index registers (the ds_bpermute form of __shfl_xor keeps six (lane ^ o) << 2 address VGPRs
// alive -- hoisted out of every loop and spilled): DPP within the 16-lane rows, v_readlane across the four rows.
template <bool MAX>
__device__ __forceinline__ float wave_reduce_dpp(float v) {
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
#define DKT_DPP(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false))
    v = op(v, DKT_DPP(v, 0xB1));        // quad_perm [1, 0, 3, 2]
    v = op(v, DKT_DPP(v, 0x4E));        // quad_perm [2, 3, 0, 1]
    v = op(v, DKT_DPP(v, 0x124));       // row_ror:4
    v = op(v, DKT_DPP(v, 0x128));       // row_ror:8
#undef DKT_DPP
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return op(op(r0, r1), op(r2, r3));
}

// The plain sweep of one diagonal tile (no interleaved work): x in the replicated column layout, all 16 pivots.
template <int P, bool LAST>
__device__ __forceinline__ void sweep_plain(float (&x)[16], float& dv, const Lane& ln, const int pn) {
    if constexpr (P < 16) {
        const float t = sweep_pivot_head<P, LAST>(x, dv, ln, pn);
        if constexpr (P + 1 < 16) sweep_rows_piece<P, P + 1, (15 - P) < 5 ? (15 - P) : 5>(x, t);
        if constexpr (P + 6 < 16) sweep_rows_piece<P, P + 6, (10 - P) < 5 ? (10 - P) : 5>(x, t);
        if constexpr (P + 11 < 16) sweep_rows_piece<P, P + 11, (5 - P) < 5 ? (5 - P) : 5>(x, t);
        sweep_plain<P + 1, LAST>(x, dv, ln, pn);
    }
}

}  // namespace dkt_mfma
