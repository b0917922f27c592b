// dkt_split.h -- shared device helpers of the episode-resident kernels that run fp32 contractions on the bf16 MFMA pipe:
// the exact 3-way bf16 split of an fp32 value, the 6-product tile update with two-level accumulation, and the
// mirrored store of a tile row of a symmetric result.  Header-only (anonymous namespace: one copy per translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include "dkt_common.h"
#include "dkt_tiles.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const float4& v, bf16x4& h, bf16x4& m, bf16x4& l) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 hi = (__bf16)x[i];
        const float r1 = x[i] - (float)hi;
        const __bf16 mi = (__bf16)r1;
        const float r2 = r1 - (float)mi;
        h[i] = hi;
        m[i] = mi;
        l[i] = (__bf16)r2;
    }
}


__device__ __forceinline__ void split3s(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}


// ---- 2-way f16 split of a SCALED fp32 value (operands known to be bounded, e.g. unit-norm rows) ----
// xs = x * 2^k (exact), h = f16(xs), m = f16(xs - h): 22 of the 24 significand bits; products hh + hm + mh rebuild the fp32
// product to 2^-21 relative (the dropped m*m term and the rounding of m), i.e. at the level of one fp32 rounding, with
// HALF the MFMAs, two thirds of the LDS traffic and 12 instead of 22 VALU instructions per float4 of the 3-way bf16 split.
// The scale keeps m out of the f16 subnormal range (which the MFMA flushes) for every element that matters and is undone
// exactly in the epilogue.
#ifndef DKT_F16_SCALE
#define DKT_F16_SCALE 32768.f          // 2^15: unit-norm operands use the f16 exponent range up to just below its maximum
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2h(const float4& v, float scale, f16x4& h, f16x4& m) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float xs = x[i] * scale;
        const _Float16 hi = (_Float16)xs;
        h[i] = hi;
#ifdef DKT_PROBE_NO_SPLIT         // measurement builds only: no low plane arithmetic
        m[i] = (_Float16)0.f;
#else
        m[i] = (_Float16)(xs - (float)hi);
#endif
    }
}

template <int NT, int RA, int RB, int SPLD, int PLANE>
__device__ __forceinline__ void sym_tiles_mfma_f16x2(f32x4* acc, const _Float16* zp, int r16, int q) {
    const _Float16* base = zp + r16 * SPLD + 8 * q;
    auto frag = [&](int plane, int blk) { return *reinterpret_cast<const f16x8*>(base + plane * PLANE + blk * 16 * SPLD); };
    auto tile = [&](f32x4& c, const f16x8& ah, const f16x8& am, int tj) {     // two-level accumulation as in the bf16 variant
        const f16x8 bh = frag(0, tj), bm = frag(1, tj);
#ifdef DKT_PROBE_ONE_PRODUCT      // measurement builds only (tools/experiments/forward_energy_probe.sh): the hh product alone -- WRONG results, the time of a third of the MFMAs
        (void)bm; (void)am;
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
#else
        f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bm, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bh, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, t, 0, 0, 0);
        c += t;
#endif
    };
    {
        const f16x8 ah = frag(0, RA), am = frag(1, RA);
#pragma unroll
        for (int tj = 0; tj <= RA; ++tj) tile(acc[tj], ah, am, tj);
    }
    if constexpr (RB >= 0) {
        const f16x8 ah = frag(0, RB), am = frag(1, RB);
#pragma unroll
        for (int tj = 0; tj <= RB; ++tj) tile(acc[RA + 1 + tj], ah, am, tj);
    }
}


// bf16 per LDS row of a plane: BK data + 16 pad -> 96 B (BK = 32) or 160 B (BK = 64): 6 / 10 sixteen-byte units,
// both == 2 mod 4: conflict-free b128 fragment reads

template <int NT, int RA, int RB, int SPLD, int PLANE>
__device__ __forceinline__ void sym_tiles_mfma_bf16x3(f32x4* acc, const __bf16* zp, int r16, int q) {
    const __bf16* base = zp + r16 * SPLD + 8 * q;
    auto frag = [&](int plane, int blk) { return *reinterpret_cast<const bf16x8*>(base + plane * PLANE + blk * 16 * SPLD); };
    // Two-level accumulation: the six products of one 32-wide slice are summed in a FRESH accumulator (smallest terms
    // first) and added to the running tile with v_add_f32.  The bf16 MFMA's final rounding truncates (measured: a
    // -3e-6 bias on a unit diagonal after 300 chained MFMAs, 10x the logp error of the round-to-nearest fp32 path);
    // against a slice-sized partial sum that bias is 50x smaller, and the fp32 add rounds to nearest.
    auto tile = [&](f32x4& c, const bf16x8& ah, const bf16x8& am, const bf16x8& al, int tj) {
        const bf16x8 bh = frag(0, tj), bm = frag(1, tj), bl = frag(2, tj);
        f32x4 t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, t, 0, 0, 0);
        c += t;
    };
    {
        const bf16x8 ah = frag(0, RA), am = frag(1, RA), al = frag(2, RA);
#pragma unroll
        for (int tj = 0; tj <= RA; ++tj) tile(acc[tj], ah, am, al, tj);
    }
    if constexpr (RB >= 0) {
        const bf16x8 ah = frag(0, RB), am = frag(1, RB), al = frag(2, RB);
#pragma unroll
        for (int tj = 0; tj <= RB; ++tj) tile(acc[RA + 1 + tj], ah, am, al, tj);
    }
}


// ---- an nn-float matrix (W, E of an episode) from global memory into LDS in ONE memory round trip ----
// Every load is issued before the first LDS store: 16-byte buffer loads at the matrix' dword alignment (an episode starts N * N floats after the
// previous one), NV = ceil(nn / 4 / NTH) per thread, the last nn % 4 floats by scalar loads.  The plain copy loop `for (i = tid; i < nn; i += NTH)
// lds[i] = src[i]` compiles to dword loads unrolled eight-fold and DRAINED (vmcnt(0)) per trip plus two remainder loops that drain every one or
// two loads: 6 - 10 dependent memory round trips per episode at N = 105 (round 5, found in the ISA of the Gram-backward prologues).
#ifdef DKT_TWINS          // DKT_LDS_STAGE_OLD=1 (twins library): the copy loop this replaced, for same-box A/B runs and the bitwise-twin test
__device__ int g_lds_stage_old = 0;
inline void lds_stage_env_sync() {
    const char* v = getenv("DKT_LDS_STAGE_OLD");
    const int f = (v && v[0] == '1') ? 1 : 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lds_stage_old), &f, sizeof(int));
}
#define DKT_LDS_STAGE_OLD_LOOP(stmt) if (g_lds_stage_old) { stmt } else
#else
inline void lds_stage_env_sync() {}
#define DKT_LDS_STAGE_OLD_LOOP(stmt)
#endif
template <int NTH, int NV>
struct LdsStage {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 t[NV];
    float ts;
    __device__ __forceinline__ void load(const float* __restrict__ src, int nn, int tid) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nn * 4, 0x00020000);
        const int nv4 = nn >> 2;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + NTH * i;
            t[i] = __builtin_amdgcn_raw_buffer_load_b128(r, v < nv4 ? 16 * v : 0x7ffffff0, 0, 0);
        }
        const int ti = 4 * nv4 + tid;
        ts = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, tid < (nn & 3) ? 4 * ti : 0x7ffffff0, 0, 0));
    }
    __device__ __forceinline__ void store(float* __restrict__ dst, int nn, int tid) const {      // dst 16-byte aligned
        const int nv4 = nn >> 2;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int v = tid + NTH * i;
            if (v < nv4) *reinterpret_cast<u32x4*>(dst + 4 * v) = t[i];
        }
        if (tid < (nn & 3)) dst[4 * nv4 + tid] = ts;
    }
};


template <int ROW>
__device__ __forceinline__ void sym_store_row(const f32x4* acc, float* Eb, int N, int r16, int q, float scale = 1.f) {
#pragma unroll
    for (int tj = 0; tj <= ROW; ++tj) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int gi = ROW * 16 + 4 * q + reg, gj = tj * 16 + r16;
            if (gi < N && gj < N && gj <= gi) {
                const float v = acc[tj][reg] * scale;
                Eb[gi * N + gj] = v;
                if (gi != gj) Eb[gj * N + gi] = v;
            }
        }
    }
}


}  // namespace
