// dkt_classkernel.hip -- the per-class element-wise maps of the non-linear base kernels and their chain rule.
//
// Reference: every class model of the one-vs-rest list owns its base kernel (one ExactGPLayer per class, methods/DKT.py:63-66,
// 352-365): RBFKernel / MaternKernel(nu = 2.5) with its own lengthscale, PolynomialKernel(power 1 / 2) with its own offset.  All C
// kernels are functions of ONE contraction per episode -- the squared distances |z_i - z_j|^2 or the Gram z_i . z_j
// (dkt_gram_f32) -- so the episode pays for one O(N^2 D) pass and
//   dkt_class_kernel_f32      base[B, NN] -> E[B, C, NN] = f(base; param_c)                     (replaces C kernel evaluations)
//   dkt_class_kernel_bwd_f32  W[B, C, N, N] = d obj / d E  ->  Wp[B, N, N] ready for dkt_gram_bwd_f32 (dZ = (Wp + Wp^T) Z)
//                             and d obj / d param [B, C]                                       (replaces autograd through them)
// run in front of / behind the ONE marginal-likelihood launch over all (episode, class) matrices (DKT_MLL_E_PER_CLASS).
// Memory-bound element-wise work: coalesced dword accesses, base read once per element for all classes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>

#include "dkt_common.h"
#include "../../include/dkt_abi.h"

namespace {

// f and f' = d f / d t of the class map at t = the map's own argument:
//   RBF       t = u = d2 / l^2            f = exp(-u / 2)                         f' = -f / 2
//   MATERN25  t = u = d2 / l^2, r = sqrt(5 max(u, 1e-30))   f = (1 + r + r^2 / 3) exp(-r)      f' = -(5 / 6)(1 + r) exp(-r)
//   POLY      t = g + offset              f = t^p (p = 1, 2)                      f' = p t^(p - 1)
template <int KIND>
__device__ __forceinline__ void class_map(const float t, const int power, float& f, float& df) {
    if constexpr (KIND == DKT_CLASSMAP_RBF) {
        f = expf(-0.5f * t);
        df = -0.5f * f;
    } else if constexpr (KIND == DKT_CLASSMAP_MATERN25) {
        const float r = sqrtf(5.0f * fmaxf(t, 1e-30f));          // gpytorch clamps d2 >= 1e-30 before the sqrt
        const float er = expf(-r);
        f = (1.0f + r + r * r * (1.0f / 3.0f)) * er;
        df = -(5.0f / 6.0f) * (1.0f + r) * er;
    } else {
        f = (power == 2) ? t * t : t;
        df = (power == 2) ? 2.0f * t : 1.0f;
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void class_kernel_fwd(const float* __restrict__ base, const float* __restrict__ param, int power,
                                                        float* __restrict__ E, int C, int NN) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= NN) return;
    const float v = base[(size_t)b * NN + k];
    float* Eb = E + (size_t)b * C * NN + k;
    for (int c = 0; c < C; ++c) {
        const float p = param[c];
        float t;
        if constexpr (KIND == DKT_CLASSMAP_POLY) t = v + p;
        else t = v / (p * p);
        float f, df;
        class_map<KIND>(t, power, f, df);
        Eb[(size_t)c * NN] = f;
    }
}

// One workgroup (8 waves) per (episode, row split); wave w owns the rows i = first + w, first + w + 8, ... of the split; lanes stride over the
// columns.  Small batches split the rows of an episode over several workgroups (a single workgroup walks its rows one load latency after
// the other: 73 us for one 105 x 105 episode, measured) and emit one parameter-gradient partial per split (dparam [B, nsplit, C]).  Per element: base once, W of every
// class once (coalesced per class).  The row sums of A (distance kinds) are wave reductions; the per-class parameter gradients are
// accumulated per thread in LDS ([C][512], no conflicts: a thread owns its slot) and reduced in a fixed order at the end.
constexpr int CKB_T = 512;

template <int KIND>
__global__ __launch_bounds__(CKB_T) void class_kernel_bwd(const float* __restrict__ W, const float* __restrict__ base,
                                                        const float* __restrict__ param, int power, float* __restrict__ Wp,
                                                        float* __restrict__ dparam, int C, int N, int nsplit) {
    extern __shared__ float dyn[];                       // [C][CKB_T] parameter-gradient partials, then [C][8] wave sums
    const int b = blockIdx.x / nsplit, sp = blockIdx.x % nsplit, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rows_per = (N + nsplit - 1) / nsplit, row0 = sp * rows_per, row1 = min(N, row0 + rows_per);
    const size_t nn = (size_t)N * N;
    const float* Wb = W + (size_t)b * C * nn;
    const float* Bb = base + (size_t)b * nn;
    float* Wpb = Wp + (size_t)b * nn;
    for (int c = 0; c < C; ++c) dyn[c * CKB_T + tid] = 0.f;
    for (int i = row0 + wave; i < row1; i += CKB_T / 64) {
        float rowsum = 0.f, adiag = 0.f;
#pragma unroll 2
        for (int j = lane; j < N; j += 64) {
            const float v = Bb[(size_t)i * N + j];
            float a = 0.f;
            for (int c = 0; c < C; ++c) {
                const float p = param[c];
                const float w = Wb[(size_t)c * nn + (size_t)i * N + j];
                float f, df;
                if constexpr (KIND == DKT_CLASSMAP_POLY) {
                    class_map<KIND>(v + p, power, f, df);
                    const float g = w * df;
                    a += g;                                              // d obj / d g_ij
                    dyn[c * CKB_T + tid] += g;                           // d obj / d offset_c
                } else {
                    const float il2 = 1.0f / (p * p);
                    const float u = v * il2;
                    class_map<KIND>(u, power, f, df);
                    const float g = w * df;                              // d obj / d u_c,ij
                    a = __builtin_fmaf(2.0f * g, il2, a);                // A = 2 d obj / d d2
                    dyn[c * CKB_T + tid] += g * (-2.0f * u / p);           // d u / d l = -2 u / l
                }
            }
            if constexpr (KIND == DKT_CLASSMAP_POLY) {
                Wpb[(size_t)i * N + j] = a;
            } else {
                rowsum += a;
                if (j == i) adiag = a;
                else Wpb[(size_t)i * N + j] = -a;
            }
        }
        if constexpr (KIND != DKT_CLASSMAP_POLY) {
            rowsum = wave_allsum(rowsum);
            adiag = wave_allsum(adiag);                                  // one lane held it
            if (lane == 0) Wpb[(size_t)i * N + i] = rowsum - adiag;     // Wp = diag(A 1) - A
        }
    }
    __syncthreads();
    for (int c = 0; c < C; ++c) {
        const float s = wave_allsum(dyn[c * CKB_T + tid]);
        __syncthreads();                                                 // (every thread has read its slot of class c)
        if (lane == 0) dyn[c * CKB_T + wave] = s;
    }
    __syncthreads();
    if (tid < C) {
        const float* d = &dyn[tid * CKB_T];
        dparam[(size_t)blockIdx.x * C + tid] = ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
    }
}

// The same chain rule for N <= 128 and C <= 8 (round 5; every 5-way episode): the kernel above walks a row element by element and class by class -- one
// 4-byte load, one exp, one LDS read-modify-write and a division per step, each class's load issued after the previous class was consumed: 0.26 of HBM at
// N = 105, C = 5.  Here a wave owns a row per trip (lanes = columns j and j + 64), issues the row's 2 (C + 1) loads at once (buffer loads; columns past N and
// classes past C through an out-of-range offset: no branch), keeps the per-class constants and the parameter-gradient partials in registers, and
// stores through the descriptor as well.  Same sums in the same order per output element as the kernel above up to the factoring of -2 u / l.
template <int KIND>
__global__ __launch_bounds__(CKB_T) void class_kernel_bwd_n128(const float* __restrict__ W, const float* __restrict__ base,
                                                             const float* __restrict__ param, int power, float* __restrict__ Wp,
                                                             float* __restrict__ dparam, int C, int N, int nsplit) {
    constexpr int CM = 8;
    __shared__ float red[CM][8];
    const int b = blockIdx.x / nsplit, sp = blockIdx.x % nsplit, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows_per = (N + nsplit - 1) / nsplit, row0 = sp * rows_per, row1 = min(N, row0 + rows_per);
    const int nn = N * N;
    typedef __amdgpu_buffer_rsrc_t brs;
    const brs wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (size_t)b * C * nn), 0, (unsigned)(C * nn * 4), 0x00020000);
    const brs br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (size_t)b * nn), 0, (unsigned)(nn * 4), 0x00020000);
    const brs pr = __builtin_amdgcn_make_buffer_rsrc(Wp + (size_t)b * nn, 0, (unsigned)(nn * 4), 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    float prm[CM], il2[CM], m2p[CM], dp[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        const float p = c < C ? param[c] : 1.0f;
        prm[c] = c < C ? p : 0.0f;                                       // POLY: offset
        il2[c] = c < C ? 1.0f / (p * p) : 0.0f;
        m2p[c] = -2.0f / p;
        dp[c] = 0.f;
    }
    const bool in0 = lane < N, in1 = lane + 64 < N;
    for (int i = row0 + wave; i < row1; i += CKB_T / 64) {
        const int o0 = in0 ? (i * N + lane) * 4 : OOB, o1 = in1 ? (i * N + lane + 64) * 4 : OOB;
        float v[2], w[CM][2];
        v[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(br, o0, 0, 0));
        v[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(br, o1, 0, 0));
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            w[c][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, c < C ? o0 : OOB, c * nn * 4, 0));
            w[c][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wr, c < C ? o1 : OOB, c * nn * 4, 0));
        }
        float a[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                float f, df;
                if constexpr (KIND == DKT_CLASSMAP_POLY) {
                    class_map<KIND>(v[e] + prm[c], power, f, df);
                    const float g = w[c][e] * df;
                    a[e] += g;
                    dp[c] += g;
                } else {
                    const float u = v[e] * il2[c];
                    class_map<KIND>(u, power, f, df);
                    const float g = w[c][e] * df;
                    a[e] = __builtin_fmaf(2.0f * g, il2[c], a[e]);
                    dp[c] = __builtin_fmaf(g, u * m2p[c], dp[c]);
                }
            }
        }
        if constexpr (KIND == DKT_CLASSMAP_POLY) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a[0]), pr, o0, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a[1]), pr, o1, 0, 0);
        } else {
            const float rowsum = wave_allsum(a[0] + a[1]);
            const float adiag = wave_allsum((lane == i ? a[0] : 0.f) + (lane + 64 == i ? a[1] : 0.f));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lane == i ? rowsum - adiag : -a[0]), pr, o0, 0, 0);      // Wp = diag(A 1) - A
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lane + 64 == i ? rowsum - adiag : -a[1]), pr, o1, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        const float s = wave_allsum(dp[c]);
        if (lane == 0) red[c][wave] = s;
    }
    __syncthreads();
    if (tid < C) {
        const float* d = red[tid];
        dparam[(size_t)blockIdx.x * C + tid] = ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
    }
}

// The same chain rule for 128 < N <= 512 (round 4; the large episodes of the one-launch per-class path, N <= 447) with 16-byte accesses and the per-class parameter
// partials in REGISTERS: the kernel above reads 4 bytes per lane and load, issues a class's load only after the previous class was consumed, and
// read-modify-writes an LDS slot per element and class -- 1.0-1.5 TB/s of the W stream (0.91 of the 3.4 ms of a 20-way rbf step of 64 episodes).
// A lane owns 4 consecutive columns of its wave's row (two such groups for N > 256); the classes go in groups of four -- four independent
// 16-byte loads in flight per lane --; classes past C read through an out-of-range offset (0: no contribution), so a group has no branches.
// A load that starts inside the row is issued whole (the row end may fall inside it; dword-aligned buffer loads): masked afterwards.
template <int KIND>
__global__ __launch_bounds__(CKB_T) void class_kernel_bwd_v4(const float* __restrict__ W, const float* __restrict__ base,
                                                           const float* __restrict__ param, int power, float* __restrict__ Wp,
                                                           float* __restrict__ dparam, int C, int N, int nsplit) {
    __shared__ float red[32][8];
    __shared__ float prm[32], prm_il2[32], prm_m2p[32];
    const int b = blockIdx.x / nsplit, sp = blockIdx.x % nsplit, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int rows_per = (N + nsplit - 1) / nsplit, row0 = sp * rows_per, row1 = min(N, row0 + rows_per);
    const size_t nn = (size_t)N * N;
    typedef __amdgpu_buffer_rsrc_t brs;
    const brs wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W + (size_t)b * C * nn), 0, (unsigned)((size_t)C * nn * 4), 0x00020000);
    const brs br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (size_t)b * nn), 0, (unsigned)(nn * 4), 0x00020000);
    const brs pr = __builtin_amdgcn_make_buffer_rsrc(Wp + (size_t)b * nn, 0, (unsigned)(nn * 4), 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    if (tid < 32) {
        const float p = tid < C ? param[tid] : 1.0f;
        prm[tid] = p;
        prm_il2[tid] = 1.0f / (p * p);
        prm_m2p[tid] = -2.0f / p;                                        // d u / d l = -2 u / l as ONE multiply per element and class (round 5: the division was
                                                                         // ~8 of the ~20 VALU instructions of an element-class step of a VALU-bound kernel)
    }
    __syncthreads();
    float dp[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) dp[c] = 0.f;
    const int nch = (N + 255) >> 8;                                        // column groups of 256 (N <= 512: 1 or 2)
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    for (int i = row0 + wave; i < row1; i += CKB_T / 64) {
        float a[2][4];
        float rowsum = 0.f, adiag = 0.f;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[ch][e] = 0.f;
            if (ch < nch) {                                                 // (uniform)
                const int j = 256 * ch + 4 * lane;
                const int off = (j < N) ? (i * N + j) * 4 : OOB;
                const u4 vb = __builtin_amdgcn_raw_buffer_load_b128(br, off, 0, 0);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(vb[e]);
#pragma unroll
                for (int c0 = 0; c0 < 32; c0 += 4) {
                    if (c0 < C) {                                           // (uniform)
                        u4 w4[4];
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc)                      // (the class offset rides in the scalar offset: c nn 4 < 2^31 for C <= 32, N <= 512)
                            w4[cc] = __builtin_amdgcn_raw_buffer_load_b128(wr, (c0 + cc < C) ? off : OOB, (int)((c0 + cc < C ? c0 + cc : 0) * nn * 4), 0);
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            const int c = c0 + cc;
                            const float p = prm[c], il2 = prm_il2[c], m2p = prm_m2p[c];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float w = (j + e < N) ? __uint_as_float(w4[cc][e]) : 0.f;       // (past the row end: the next row's values)
                                float f, df;
                                if constexpr (KIND == DKT_CLASSMAP_POLY) {
                                    class_map<KIND>(v[e] + p, power, f, df);
                                    const float g = w * df;
                                    a[ch][e] += g;                          // d obj / d g_ij
                                    dp[c] += g;                             // d obj / d offset_c
                                } else {
                                    const float u = v[e] * il2;
                                    class_map<KIND>(u, power, f, df);
                                    const float g = w * df;                 // d obj / d u_c,ij
                                    a[ch][e] = __builtin_fmaf(2.0f * g, il2, a[ch][e]);             // A = 2 d obj / d d2
                                    dp[c] = __builtin_fmaf(g, u * m2p, dp[c]);                      // d u / d l = -2 u / l
                                }
                            }
                        }
                    }
                }
                if constexpr (KIND != DKT_CLASSMAP_POLY) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        rowsum += a[ch][e];
                        if (j + e == i) adiag = a[ch][e];
                    }
                }
            }
        }
        if constexpr (KIND != DKT_CLASSMAP_POLY) {
            rowsum = wave_allsum(rowsum);
            adiag = wave_allsum(adiag);                                      // one lane held it
        }
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (ch < nch) {
                const int j = 256 * ch + 4 * lane;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (KIND == DKT_CLASSMAP_POLY) o[e] = a[ch][e];
                    else o[e] = (j + e == i) ? rowsum - adiag : -a[ch][e];   // Wp = diag(A 1) - A
                }
                if (j + 3 < N) {
                    const u4 ov = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(ov, pr, (i * N + j) * 4, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[e]), pr, (j + e < N) ? (i * N + j + e) * 4 : OOB, 0, 0);
                }
            }
        }
    }
    // parameter partials: wave sums -> a fixed-order sum over the 8 waves (deterministic)
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        if (c < C) {
            const float s = wave_allsum(dp[c]);
            if (lane == 0) red[c][wave] = s;
        }
    }
    __syncthreads();
    if (tid < C) {
        const float* d = red[tid];
        dparam[(size_t)blockIdx.x * C + tid] = ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
    }
}

}  // namespace

extern "C" int dkt_class_kernel_f32(const float* base, int kind, const float* param, int power, float* E, int B, int C, int NN,
                                    void* stream) {
    if (!base || !param || !E || B <= 0 || C <= 0 || NN <= 0) return DKT_ERR_BAD_ARG;
    if (kind == DKT_CLASSMAP_POLY && power != 1 && power != 2) return DKT_ERR_BAD_ARG;
    if (B > 65535) return DKT_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((NN + 255) / 256, B), block(256);
    switch (kind) {
        case DKT_CLASSMAP_RBF: hipLaunchKernelGGL(class_kernel_fwd<DKT_CLASSMAP_RBF>, grid, block, dkt_lds_pad("DKT_PAD_CK_FWD"), st, base, param, power, E, C, NN); break;
        case DKT_CLASSMAP_MATERN25: hipLaunchKernelGGL(class_kernel_fwd<DKT_CLASSMAP_MATERN25>, grid, block, dkt_lds_pad("DKT_PAD_CK_FWD"), st, base, param, power, E, C, NN); break;
        case DKT_CLASSMAP_POLY: hipLaunchKernelGGL(class_kernel_fwd<DKT_CLASSMAP_POLY>, grid, block, dkt_lds_pad("DKT_PAD_CK_FWD"), st, base, param, power, E, C, NN); break;
        default: return DKT_ERR_BAD_ARG;
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

static int g_ck_v4 = -1, g_ck_n128 = -1;
void dkt_classkernel_reload_env() { g_ck_v4 = -1; g_ck_n128 = -1; }     // dkt_reload_env()
static bool class_bwd_n128() {                                         // DKT_CLASS_BWD_N128=0 (twins library): the dword kernel for N <= 128 too (A/B, twin test)
    if (g_ck_n128 < 0) { const char* v = dkt_variant_env("DKT_CLASS_BWD_N128"); g_ck_n128 = (v && v[0] == '0') ? 0 : 1; }
    return g_ck_n128 != 0;
}

extern "C" int dkt_class_kernel_bwd_nsplit(int B, int N) {
    if (B <= 0 || N <= 0) return 1;
    int ns = 1;
    while (ns < 16 && B * ns < 256 && N >= 16 * ns) ns *= 2;             // enough workgroups to cover the CUs, >= 8 rows each
    return ns;
}

extern "C" int dkt_class_kernel_bwd_f32(const float* W, const float* base, int kind, const float* param, int power, float* Wp,
                                        float* dparam, int B, int C, int N, void* stream) {
    if (!W || !base || !param || !Wp || !dparam || B <= 0 || C <= 0 || N <= 0) return DKT_ERR_BAD_ARG;
    const int nsplit = dkt_class_kernel_bwd_nsplit(B, N);
    if (kind == DKT_CLASSMAP_POLY && power != 1 && power != 2) return DKT_ERR_BAD_ARG;
    if (C > 32) return DKT_ERR_TOO_LARGE;                                // 32 x 512 floats of LDS partials (64 KB)
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B * nsplit), block(CKB_T);
    if (g_ck_v4 < 0) { const char* v = dkt_variant_env("DKT_CLASS_BWD_V4"); g_ck_v4 = (v && v[0] == '0') ? 0 : 1; }      // (0: the dword kernel, a measurement twin)
    // round 5: all loads of a row in flight, constants and partials in registers.  Same-box A/B (tools/class_bwd_ab.py, profiles/r05/v12_class_bwd_ab.log; 2048 episodes,
    // C = 5): N = 105 rbf 0.294 -> 0.139 ms (0.27 -> 0.57 of HBM), matern 0.351 -> 0.214, poly 0.241 -> 0.132, N = 80 0.204 -> 0.096; C = 8, N = 128: 0.213 -> 0.145;
    // N = 25 (a row fills 25 of a wave's 128 column slots) 0.092 -> 0.128: from 65 rows.
    if (N > 64 && N <= 128 && C <= 8 && class_bwd_n128()) {
        switch (kind) {
            case DKT_CLASSMAP_RBF: hipLaunchKernelGGL(class_kernel_bwd_n128<DKT_CLASSMAP_RBF>, grid, block, dkt_lds_pad("DKT_PAD_CK_BWD"), st, W, base, param, power, Wp, dparam, C, N, nsplit); break;
            case DKT_CLASSMAP_MATERN25: hipLaunchKernelGGL(class_kernel_bwd_n128<DKT_CLASSMAP_MATERN25>, grid, block, dkt_lds_pad("DKT_PAD_CK_BWD"), st, W, base, param, power, Wp, dparam, C, N, nsplit); break;
            case DKT_CLASSMAP_POLY: hipLaunchKernelGGL(class_kernel_bwd_n128<DKT_CLASSMAP_POLY>, grid, block, dkt_lds_pad("DKT_PAD_CK_BWD"), st, W, base, param, power, Wp, dparam, C, N, nsplit); break;
            default: return DKT_ERR_BAD_ARG;
        }
        return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    }
    // (N <= 128: a row of 4-column groups leaves more than half of a wave's lanes idle -- 0.58 vs 0.28 ms at N = 105, C = 5, 2048 episodes: the dword kernel stays for C > 8)
    if (N > 128 && N <= 512 && g_ck_v4 != 0) {
        switch (kind) {
            case DKT_CLASSMAP_RBF: hipLaunchKernelGGL(class_kernel_bwd_v4<DKT_CLASSMAP_RBF>, grid, block, dkt_lds_pad("DKT_PAD_CK_BWD"), st, W, base, param, power, Wp, dparam, C, N, nsplit); break;
            case DKT_CLASSMAP_MATERN25: hipLaunchKernelGGL(class_kernel_bwd_v4<DKT_CLASSMAP_MATERN25>, grid, block, dkt_lds_pad("DKT_PAD_CK_BWD"), st, W, base, param, power, Wp, dparam, C, N, nsplit); break;
            case DKT_CLASSMAP_POLY: hipLaunchKernelGGL(class_kernel_bwd_v4<DKT_CLASSMAP_POLY>, grid, block, dkt_lds_pad("DKT_PAD_CK_BWD"), st, W, base, param, power, Wp, dparam, C, N, nsplit); break;
            default: return DKT_ERR_BAD_ARG;
        }
        return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    }
    const size_t lds = (size_t)C * CKB_T * sizeof(float);
    switch (kind) {
        case DKT_CLASSMAP_RBF: hipLaunchKernelGGL(class_kernel_bwd<DKT_CLASSMAP_RBF>, grid, block, lds, st, W, base, param, power, Wp, dparam, C, N, nsplit); break;
        case DKT_CLASSMAP_MATERN25: hipLaunchKernelGGL(class_kernel_bwd<DKT_CLASSMAP_MATERN25>, grid, block, lds, st, W, base, param, power, Wp, dparam, C, N, nsplit); break;
        case DKT_CLASSMAP_POLY: hipLaunchKernelGGL(class_kernel_bwd<DKT_CLASSMAP_POLY>, grid, block, lds, st, W, base, param, power, Wp, dparam, C, N, nsplit); break;
        default: return DKT_ERR_BAD_ARG;
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
