// dkt_h2_tiles.h -- 16 x 16 tile primitives on v_mfma_f32_16x16x16_f16 with scaled 2-way f16 splits (dkt_mll_h2.hip; exercised in
// isolation by dkt_diag.hip / test_h2_tile_primitives).  A tile in the MFMA accumulator layout (lane (g, c), register q  <->  element
// [4g + q][c]) becomes, with its four registers converted to f16 and packed into two VGPRs, a legal A operand (A = X^T) and a legal
// B operand of that instruction: D += X^T Y is one instruction per pair of planes.  A split tile is (h, m) = (f16(x s), f16(x s - h)).
#pragma once
#include "dkt_mfma_tiles.h"

namespace dkt_mfma {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
struct Sp { h4 h, m; };          // a split tile: 4 VGPRs

constexpr float TWO30 = 1073741824.0f, TWOM30 = 9.313225746154785e-10f, TWOM15 = 3.0517578125e-05f;

// x * scale -> (h, m), packed: h = f16(x s) (the product is exact, one rounding to nearest), m = f16(x s - h) (the difference is exact in
// fp32).  10 VALU per tile: 4 v_mul_f32, 2 v_cvt_pk_f16_f32, 4 v_fma_mix{lo,hi}_f16 (fp32 FMA with the f16 half of h as addend).
// Deliberately plain C, not inline asm: the MFMA <-> VALU wait states (an MFMA result read by the next VALU instruction, a VALU write
// into a register an MFMA in flight still reads) are inserted by the compiler's hazard recogniser, which does not look inside asm
// statements -- an 8-instruction asm version of this function read accumulators 2 instructions after the MFMA that produced them.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 split_h2(const f32x4 v, const float scale) {
    const h2v h01 = __builtin_convertvector((f32x2){v[0] * scale, v[1] * scale}, h2v), h23 = __builtin_convertvector((f32x2){v[2] * scale, v[3] * scale}, h2v);
    h2v m01, m23;
    m01[0] = (_Float16)__builtin_fmaf(v[0], scale, -(float)h01[0]); m01[1] = (_Float16)__builtin_fmaf(v[1], scale, -(float)h01[1]);
    m23[0] = (_Float16)__builtin_fmaf(v[2], scale, -(float)h23[0]); m23[1] = (_Float16)__builtin_fmaf(v[3], scale, -(float)h23[1]);
    return (f32x4){__builtin_bit_cast(float, h01), __builtin_bit_cast(float, h23), __builtin_bit_cast(float, m01), __builtin_bit_cast(float, m23)};
}
// (h, m) -> h + m as fp32 (exact): one v_fma_mix_f32 per element
__device__ __forceinline__ f32x4 join_h2(const f32x4 s) {
    const Sp sp = __builtin_bit_cast(Sp, s);        // (of the whole vector: __builtin_bit_cast of an ext_vector ELEMENT reads element 0 whatever the index)
    return (f32x4){__builtin_fmaf((float)sp.h[0], 1.0f, (float)sp.m[0]), __builtin_fmaf((float)sp.h[1], 1.0f, (float)sp.m[1]),
                   __builtin_fmaf((float)sp.h[2], 1.0f, (float)sp.m[2]), __builtin_fmaf((float)sp.h[3], 1.0f, (float)sp.m[3])};
}

__device__ __forceinline__ Sp as_sp(const f32x4 r) { return __builtin_bit_cast(Sp, r); }

// one of the three plane products of D += X^T Y on split tiles: 0: h m, 1: m h, 2: h h (small terms first)
template <int WHICH>
__device__ __forceinline__ f32x4 xtyh1(const f32x4 xr, const f32x4 yr, const f32x4 c) {
    const Sp x = as_sp(xr), y = as_sp(yr);
    if constexpr (WHICH == 0) return __builtin_amdgcn_mfma_f32_16x16x16f16(x.h, y.m, c, 0, 0, 0);
    else if constexpr (WHICH == 1) return __builtin_amdgcn_mfma_f32_16x16x16f16(x.m, y.h, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x16f16(x.h, y.h, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 xtyh(const f32x4 x, const f32x4 y, f32x4 c) {
    c = xtyh1<0>(x, y, c);
    c = xtyh1<1>(x, y, c);
    return xtyh1<2>(x, y, c);
}
__device__ __forceinline__ f32x4 xtyh0(const f32x4 x, const f32x4 y) { return xtyh(x, y, (f32x4){0.f, 0.f, 0.f, 0.f}); }
// two independent chains advanced alternately
__device__ __forceinline__ void xtyh2(const f32x4 xa, const f32x4 ya, f32x4& ca, const f32x4 xb, const f32x4 yb, f32x4& cb) {
    ca = xtyh1<0>(xa, ya, ca); cb = xtyh1<0>(xb, yb, cb);
    ca = xtyh1<1>(xa, ya, ca); cb = xtyh1<1>(xb, yb, cb);
    ca = xtyh1<2>(xa, ya, ca); cb = xtyh1<2>(xb, yb, cb);
}
__device__ __forceinline__ void xty2(const f32x4 xa, const f32x4 ya, f32x4& ca, const f32x4 xb, const f32x4 yb, f32x4& cb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ca = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[q], ya[q], ca, 0, 0, 0);
        cb = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[q], yb[q], cb, 0, 0, 0);
    }
}

// -X^T of a split tile, split again: every plane goes through the matrix pipe against -I (one non-zero term per element: exact)
__device__ __forceinline__ f32x4 neg_transpose_h2(const f32x4 xr, const h4 negI) {
    const Sp x = as_sp(xr);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 th = __builtin_amdgcn_mfma_f32_16x16x16f16(x.h, negI, z, 0, 0, 0);
    const f32x4 tm = __builtin_amdgcn_mfma_f32_16x16x16f16(x.m, negI, z, 0, 0, 0);
    typedef __fp16 hh2 __attribute__((ext_vector_type(2)));
    const hh2 a0 = __builtin_amdgcn_cvt_pkrtz(th[0], th[1]), a1 = __builtin_amdgcn_cvt_pkrtz(th[2], th[3]);
    const hh2 b0 = __builtin_amdgcn_cvt_pkrtz(tm[0], tm[1]), b1 = __builtin_amdgcn_cvt_pkrtz(tm[2], tm[3]);
    return (f32x4){__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1), __builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1)};
}

// 2^(15 - e) and its reciprocal for a positive finite x = f 2^e, 0.5 <= f < 1: the scale that maps [0, x] into [0, 2^15)
__device__ __forceinline__ float scale_for(const float xmax, float& inv) {
    const unsigned be = (__float_as_uint(xmax) >> 23) & 0xffu;            // biased exponent of xmax; e = be - 126
    inv = __uint_as_float(((be - 14u) & 0xffu) << 23);                      // 2^(e - 15)
    return __uint_as_float(((268u - be) & 0xffu) << 23);                    // 2^(15 - e)
}

}  // namespace dkt_mfma
