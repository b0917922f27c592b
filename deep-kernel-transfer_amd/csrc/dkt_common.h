// Shared device helpers for the DKT hot-path kernels (gfx950 / CDNA4 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DKT_WAVE 64

// Measurement / validation switches.  The PRODUCT library (libdkt_hip.so) has none of them: every variant kernel and every switch
// that selects one is compiled out, the defaults are the product.  The twins library (libdkt_twins.so: the same sources with -DDKT_TWINS, loaded by the
// tests and the A/B tools only) reads them from the environment.
#include <cstdlib>
static inline const char* dkt_variant_env(const char* name) {
#ifdef DKT_TWINS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, DKT_WAVE);
    return v;  // valid in lane 0
}

__device__ __forceinline__ float wave_allsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, DKT_WAVE);
    return v;
}

// Sum over a 256-thread block; result returned to every thread. `red` = >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// 16-byte global load of 4 consecutive floats of a row, zero-filled outside [0, ncols).
// vec_ok: row base and ncols are 16-byte compatible (ncols % 4 == 0, base aligned).
__device__ __forceinline__ float4 load4_guard(const float* row, int col, int ncols, bool row_ok, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok) return v;
    if (vec_ok) {
        if (col < ncols) v = *reinterpret_cast<const float4*>(row + col);
    } else {
        if (col + 0 < ncols) v.x = row[col + 0];
        if (col + 1 < ncols) v.y = row[col + 1];
        if (col + 2 < ncols) v.z = row[col + 2];
        if (col + 3 < ncols) v.w = row[col + 3];
    }
    return v;
}

// Twins library only: bytes of dynamic LDS a launch reserves without touching them -- caps the workgroups per CU for occupancy A/Bs of the streaming kernels
// (tools/occupancy_pad_ab.py; the product returns 0).
static inline unsigned dkt_lds_pad(const char* name) {
    const char* v = dkt_variant_env(name);
    return v ? (unsigned)atoi(v) : 0u;
}
