// dkt_gram.hip -- base-kernel (Gram) matrix build and its backward, fp32-exact MFMA (gfx950).
//
//   gram_nt_kernel : E[b] = k(A[b], Bm[b]),  A:[M,D], Bm:[N,D]  (NT contraction over D)
//   gram_bwd_kernel: dZ[b] = s_b (W[b] + W[b]^T) Z[b]           (NN contraction over N)
//
// Both use v_mfma_f32_16x16x4_f32 (exact fp32, 32-cycle issue): a 64x64 output tile per 256-thread
// workgroup, 2x2 waves, each wave 2x2 MFMA fragments.  Operand tiles are staged global -> registers
// -> LDS (double buffered, one barrier per K step) with 16-byte loads; fragment reads are
// ds_read_b128 with a k-permutation (lane group q holds k = 4q..4q+3 of each 16-wide K slice, the
// t-th MFMA of the slice contracts k = {t, 4+t, 8+t, 12+t}), row stride 40 floats = conflict-free
// for the b128 lane groups.
//
// Replaces ExactGPLayer.forward -> covar_module(x) (reference methods/DKT.py:375-378,
// methods/DKT_regression.py:126-129) and autograd through it (DKT.py:163).
#include "dkt_common.h"
#include <cstdlib>
#include "../../include/dkt_abi.h"

bool dkt_gram_sym_ep_launch(const float* Z, float* E, int B, int N, int D, bool unit, hipStream_t st);
bool dkt_gram_bwd_ep_launch(const float* W, const float* Z, float* dZ, int B, int N, int D, const float* sc, bool unit, hipStream_t st);
bool dkt_gram_sym_big_launch(const float* Z, float* E, int B, int N, int D, bool unit, hipStream_t st);
bool dkt_gram_fewep_applies(int B, int N, int D);                                  // dkt_gram_ep.hip: fewer episodes than the episode-resident kernels take
// wave-per-episode kernels for N <= 32 (dkt_gram_small.hip): every kind, symmetric; DKT_GRAM_SMALL=0 keeps the generic kernels
bool dkt_gram_small_launch(const float* Z, float* E, int B, int N, int D, int kind, const float* lengthscale, hipStream_t st);
bool dkt_gram_dist_ep_launch(const float* Z, float* E, int B, int N, int D, int kind, const float* lengthscale, hipStream_t st);      // dkt_frontend.hip
bool dkt_gram_small_bwd_launch(const float* W, const float* Z, float* dZ, int B, int N, int D, const float* sc, hipStream_t st);
bool dkt_gram_bwd_big_launch(const float* W, const float* Z, float* dZ, int B, int N, int D, const float* sc, unsigned flags, hipStream_t st);

namespace {

constexpr int GT = 64;         // output tile edge
constexpr int GBK = 32;        // K per stage
constexpr int GLD = GBK + 8;   // LDS row stride (floats) of a [64][GBK] operand tile
constexpr int BLD = GT + 4;    // LDS row stride of the [GBK][64] NN operand tile

__device__ __forceinline__ void mfma4(f32x4& acc, const f32x4& a, const f32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// E = k(A, Bm).  grid = (tiles over N, tiles over M, B).  SYM: Bm == A, upper tiles skipped and
// the lower tiles mirrored on store.  KIND 1 (RBF): operands are shifted by row 0 of A (any common
// shift leaves |a-b| unchanged; it removes the shared offset of ReLU features the way GPyTorch's
// mean-centring does), squared row norms are accumulated while staging, and the epilogue forms
// d2 = |a|^2 + |b|^2 - 2 a.b (clamped at 0) and E = exp(-0.5 d2 / l^2).
// ------------------------------------------------------------------------------------------
template <int KIND, bool SYM>
__global__ __launch_bounds__(256) void gram_nt_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                      float* __restrict__ E, int M, int N, int D,
                                                      const float* __restrict__ lengthscale) {
    const int tn = blockIdx.x, tm = blockIdx.y, b = blockIdx.z;
    if (SYM && tn > tm) return;
    const bool diag = SYM && (tn == tm);
    const float* Ab = A + (size_t)b * M * D;
    const float* Bb = SYM ? Ab : Bm + (size_t)b * N * D;
    float* Eb = E + (size_t)b * M * N;
    const int m0 = tm * GT, n0 = tn * GT;

    __shared__ __attribute__((aligned(16))) float As[2][GT * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GT * GLD];
    __shared__ float nrmA[GT], nrmB[GT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, q = lane >> 4;
    const int lr = tid >> 3;          // staging row 0..31 (and +32)
    const int lc = (tid & 7) * 4;     // staging k offset within the stage
    const bool vec_ok = ((D & 3) == 0) && ((((uintptr_t)Ab) & 15) == 0) && ((((uintptr_t)Bb) & 15) == 0);

    float4 ra[2], rb[2];
    float na[2] = {0.f, 0.f}, nb[2] = {0.f, 0.f};

    auto gload = [&](int k0) {
        const int k = k0 + lc;
        float4 ref = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KIND != DKT_KERNEL_LINEAR) ref = load4_guard(Ab, k, D, true, vec_ok);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rowa = m0 + lr + 32 * h;
            const bool oka = rowa < M;
            ra[h] = load4_guard(Ab + (size_t)rowa * D, k, D, oka, vec_ok);
            if (KIND != DKT_KERNEL_LINEAR && oka) {
                if (k + 0 < D) ra[h].x -= ref.x;
                if (k + 1 < D) ra[h].y -= ref.y;
                if (k + 2 < D) ra[h].z -= ref.z;
                if (k + 3 < D) ra[h].w -= ref.w;
                na[h] += ra[h].x * ra[h].x + ra[h].y * ra[h].y + ra[h].z * ra[h].z + ra[h].w * ra[h].w;
            }
            if (!diag) {
                const int rowb = n0 + lr + 32 * h;
                const bool okb = rowb < N;
                rb[h] = load4_guard(Bb + (size_t)rowb * D, k, D, okb, vec_ok);
                if (KIND != DKT_KERNEL_LINEAR && okb) {
                    if (k + 0 < D) rb[h].x -= ref.x;
                    if (k + 1 < D) rb[h].y -= ref.y;
                    if (k + 2 < D) rb[h].z -= ref.z;
                    if (k + 3 < D) rb[h].w -= ref.w;
                    nb[h] += rb[h].x * rb[h].x + rb[h].y * rb[h].y + rb[h].z * rb[h].z + rb[h].w * rb[h].w;
                }
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(&As[buf][(lr + 32 * h) * GLD + lc]) = ra[h];
            if (!diag) *reinterpret_cast<float4*>(&Bs[buf][(lr + 32 * h) * GLD + lc]) = rb[h];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (D + GBK - 1) / GBK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * GBK);
        const float* as = As[buf];
        const float* bs = diag ? As[buf] : Bs[buf];
#pragma unroll
        for (int kk = 0; kk < GBK / 16; ++kk) {
            const int ko = kk * 16 + 4 * q;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(&as[(wm * 32 + r16) * GLD + ko]);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(&as[(wm * 32 + 16 + r16) * GLD + ko]);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(&bs[(wn * 32 + r16) * GLD + ko]);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(&bs[(wn * 32 + 16 + r16) * GLD + ko]);
            mfma4(acc[0][0], a0, b0);
            mfma4(acc[0][1], a0, b1);
            mfma4(acc[1][0], a1, b0);
            mfma4(acc[1][1], a1, b1);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    float inv_l2 = 0.f;
    if (KIND != DKT_KERNEL_LINEAR) {
        // reduce the per-thread partial norms over the 8 threads sharing a staging row
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float va = na[h], vb = nb[h];
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                va += __shfl_xor(va, o, DKT_WAVE);
                vb += __shfl_xor(vb, o, DKT_WAVE);
            }
            if ((tid & 7) == 0) {
                nrmA[lr + 32 * h] = va;
                nrmB[lr + 32 * h] = diag ? va : vb;
            }
        }
        __syncthreads();
        const float l = lengthscale[0];
        inv_l2 = 1.0f / (l * l);
    }

    // epilogue: C/D layout of 16x16 MFMA: row = 4*(lane>>4) + reg, col = lane & 15
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int fj = 0; fj < 2; ++fj)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int lm = wm * 32 + fi * 16 + 4 * q + reg;
                const int ln = wn * 32 + fj * 16 + r16;
                const int gm = m0 + lm, gn = n0 + ln;
                if (gm >= M || gn >= N) continue;
                float v = acc[fi][fj][reg];
                if (KIND != DKT_KERNEL_LINEAR) {
                    float d2 = nrmA[lm] + nrmB[ln] - 2.0f * v;
                    d2 = d2 > 0.f ? d2 : 0.f;
                    if (SYM && gm == gn) d2 = 0.f;
                    v = (KIND == DKT_KERNEL_RBF) ? expf(-0.5f * d2 * inv_l2) : d2 * inv_l2;
                }
                if (SYM) {
                    if (diag && gn > gm) continue;  // keep the matrix exactly symmetric
                    Eb[(size_t)gm * N + gn] = v;
                    if (gm != gn) Eb[(size_t)gn * N + gm] = v;
                } else {
                    Eb[(size_t)gm * N + gn] = v;
                }
            }
}

// ------------------------------------------------------------------------------------------
// dZ[b] = s_b (W + W^T) Z.   grid = (tiles over D, tiles over N, B).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_bwd_kernel(const float* __restrict__ W, const float* __restrict__ Z,
                                                       float* __restrict__ dZ, int N, int D,
                                                       const float* __restrict__ ep_scale) {
    const int td = blockIdx.x, tm = blockIdx.y, b = blockIdx.z;
    const float* Wb = W + (size_t)b * N * N;
    const float* Zb = Z + (size_t)b * N * D;
    float* dZb = dZ + (size_t)b * N * D;
    const int m0 = tm * GT, d0 = td * GT;

    __shared__ __attribute__((aligned(16))) float As[2][GT * GLD];   // [i][k]  (W + W^T)
    __shared__ __attribute__((aligned(16))) float Bs[2][GBK * BLD];  // [k][d]  Z

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, q = lane >> 4;
    const int lr = tid >> 3, lc = (tid & 7) * 4;    // A staging: rows lr, lr+32; k offset lc
    const int kr = tid >> 4, dc = (tid & 15) * 4;   // B staging: k rows kr, kr+16; d offset dc
    const bool vec_ok = ((D & 3) == 0) && ((((uintptr_t)Zb) & 15) == 0);

    float4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int i = m0 + lr + 32 * h;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + lc + e;
                v[e] = (i < N && k < N) ? (Wb[(size_t)i * N + k] + Wb[(size_t)k * N + i]) : 0.f;
            }
            ra[h] = make_float4(v[0], v[1], v[2], v[3]);
            const int k = k0 + kr + 16 * h;
            rb[h] = load4_guard(Zb + (size_t)k * D, d0 + dc, D, k < N, vec_ok);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(&As[buf][(lr + 32 * h) * GLD + lc]) = ra[h];
            *reinterpret_cast<float4*>(&Bs[buf][(kr + 16 * h) * BLD + dc]) = rb[h];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (N + GBK - 1) / GBK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * GBK);
        const float* as = As[buf];
        const float* bs = Bs[buf];
#pragma unroll
        for (int kk = 0; kk < GBK / 16; ++kk) {
            const int ko = kk * 16 + 4 * q;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(&as[(wm * 32 + r16) * GLD + ko]);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(&as[(wm * 32 + 16 + r16) * GLD + ko]);
            f32x4 b0, b1;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                b0[t] = bs[(ko + t) * BLD + wn * 32 + r16];
                b1[t] = bs[(ko + t) * BLD + wn * 32 + 16 + r16];
            }
            mfma4(acc[0][0], a0, b0);
            mfma4(acc[0][1], a0, b1);
            mfma4(acc[1][0], a1, b0);
            mfma4(acc[1][1], a1, b1);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }

    const float s = ep_scale ? ep_scale[b] : 1.0f;
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int fj = 0; fj < 2; ++fj)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gm = m0 + wm * 32 + fi * 16 + 4 * q + reg;
                const int gd = d0 + wn * 32 + fj * 16 + r16;
                if (gm < N && gd < D) dZb[(size_t)gm * D + gd] = s * acc[fi][fj][reg];
            }
}

// ------------------------------------------------------------------------------------------
// RBF chain rule: Wp = diag(A 1) - A, A = -(Ws o E)/l^2;  dl = sum Ws E d2 / l^3.
// one workgroup per episode, one row per thread (strided).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rbf_bwd_kernel(const float* __restrict__ W, const float* __restrict__ E,
                                                      const float* __restrict__ lengthscale, float* __restrict__ Wp,
                                                      float* __restrict__ dl, int N) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* Wb = W + (size_t)b * N * N;
    const float* Eb = E + (size_t)b * N * N;
    float* Wpb = Wp + (size_t)b * N * N;
    const float l = lengthscale[0];
    const float inv_l2 = 1.0f / (l * l);
    float dl_part = 0.f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float rowsum = 0.f;
        for (int j = 0; j < N; ++j) {
            const float ws = 0.5f * (Wb[(size_t)i * N + j] + Wb[(size_t)j * N + i]);
            const float e = Eb[(size_t)i * N + j];
            const float a = -(ws * e) * inv_l2;
            rowsum += a;
            if (j != i) Wpb[(size_t)i * N + j] = -a;
            if (e > 0.f) dl_part += ws * e * (-2.0f * logf(e));  // = ws e d2 / l^2
        }
        const float wsd = Wb[(size_t)i * N + i];
        const float ad = -(wsd * Eb[(size_t)i * N + i]) * inv_l2;
        Wpb[(size_t)i * N + i] = rowsum - ad;
    }
    const float tot = block_sum_256(dl_part, red);
    if (threadIdx.x == 0) dl[b] = tot / l;
}

// scaled squared distance U = d2 / l^2: chain rule.  dU_ij = (2 / l^2)(z_i - z_j).(dz_i - dz_j), so with
// A = 2 Ws / l^2 and Wp = diag(A 1) - A the gradient is dZ = 2 Wp Z (= dkt_gram_bwd_f32(Wp, Z));  dl = sum Ws U (-2 / l).
__global__ __launch_bounds__(256) void sqdist_bwd_kernel(const float* __restrict__ W, const float* __restrict__ U,
                                                         const float* __restrict__ lengthscale, float* __restrict__ Wp,
                                                         float* __restrict__ dl, int N) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const float* Wb = W + (size_t)b * N * N;
    const float* Ub = U + (size_t)b * N * N;
    float* Wpb = Wp + (size_t)b * N * N;
    const float l = lengthscale[0];
    const float inv_l2 = 1.0f / (l * l);
    float dl_part = 0.f;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        float rowsum = 0.f;
        for (int j = 0; j < N; ++j) {
            const float ws = 0.5f * (Wb[(size_t)i * N + j] + Wb[(size_t)j * N + i]);
            const float a = 2.0f * ws * inv_l2;
            rowsum += a;
            if (j != i) Wpb[(size_t)i * N + j] = -a;
            dl_part += ws * Ub[(size_t)i * N + j];
        }
        Wpb[(size_t)i * N + i] = rowsum - 2.0f * Wb[(size_t)i * N + i] * inv_l2;
    }
    const float tot = block_sum_256(dl_part, red);
    if (threadIdx.x == 0) dl[b] = -2.0f * tot / l;
}


// ------------------------------------------------------------------------------------------
// A handful of episodes (the reference's literal loop is ONE episode per step, DKT.py:117-164): every kernel above gives an episode to one to three workgroups,
// whose K loop is then a chain of dependent memory round trips (50 steps at D = 1600: 57 us for one 105 x 1600 episode, on 3 of 256 CUs).  Here a workgroup owns
// ONE 16 x 16 tile of the lower block triangle and its 8 waves each contract every eighth 16-feature group, straight from global memory into the MFMA operand
// layout (lane (r, q) loads Z[16 i + r][16 s + 4 q .. + 3]: the t-th MFMA of the group contracts k = {t, 4 + t, 8 + t, 12 + t} on both operands); the partial tiles
// meet in LDS in wave order (bitwise reproducible).  Z is re-read once per tile column from L2 -- irrelevant at this size.  Exact fp32 (v_mfma_f32_16x16x4_f32).
// ------------------------------------------------------------------------------------------
constexpr int FEW_WAVES = 8, FEW_UNROLL = 4;
__global__ __launch_bounds__(64 * FEW_WAVES) void gram_sym_fewep_kernel(const float* __restrict__ Z, float* __restrict__ E, const int N, const int D) {
    __shared__ __attribute__((aligned(16))) float part[FEW_WAVES][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    int ti = 0;
    {
        int t = blockIdx.x;                          // tile t of the lower block triangle, row by row
        while (t > ti) { t -= ti + 1; ++ti; }
        // (ti, t) = (row tile, column tile)
        const int tj = t;
        const float* Zb = Z + (size_t)blockIdx.y * N * D;
        const int ra = 16 * ti + r16, rb = 16 * tj + r16;
        const float* pa = Zb + (size_t)ra * D + 4 * q;
        const float* pb = Zb + (size_t)rb * D + 4 * q;
        const bool oka = ra < N, okb = rb < N, diag = ti == tj;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int S = (D + 15) >> 4;
        for (int s0 = wave; s0 < S; s0 += FEW_WAVES * FEW_UNROLL) {
            float4 a[FEW_UNROLL], b[FEW_UNROLL];
#pragma unroll
            for (int u = 0; u < FEW_UNROLL; ++u) {
                const int s = s0 + u * FEW_WAVES, k = 16 * s + 4 * q;
                const bool in = s < S && k < D;
                a[u] = (in && oka) ? *reinterpret_cast<const float4*>(pa + 16 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
                b[u] = diag ? a[u] : ((in && okb) ? *reinterpret_cast<const float4*>(pb + 16 * s) : make_float4(0.f, 0.f, 0.f, 0.f));
            }
#pragma unroll
            for (int u = 0; u < FEW_UNROLL; ++u) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
            }
        }
        *reinterpret_cast<f32x4*>(&part[wave][lane * 4]) = acc;
        __syncthreads();
        if (tid < 256) {
            // thread = element (row = tid / 16, col = tid % 16) of the tile: accumulator lane (row / 4, col), register row % 4
            const int row = tid >> 4, col = tid & 15, at = ((row >> 2) * 16 + col) * 4 + (row & 3);
            float v = part[0][at];
#pragma unroll
            for (int w = 1; w < FEW_WAVES; ++w) v += part[w][at];
            const int gi = 16 * ti + row, gj = 16 * tj + col;
            float* Eb = E + (size_t)blockIdx.y * N * N;
            if (gi < N && gj < N) {
                Eb[(size_t)gi * N + gj] = v;                             // (a diagonal tile is bitwise symmetric: both halves sum the same products in the same order)
                if (!diag) Eb[(size_t)gj * N + gi] = v;
            }
        }
    }
}

}  // namespace

extern "C" int dkt_sqdist_bwd_f32(const float* W, const float* U, const float* lengthscale, float* Wp,
                                  float* dlengthscale, int B, int N, void* stream) {
    if (!W || !U || !lengthscale || !Wp || !dlengthscale || B <= 0 || N <= 0) return DKT_ERR_BAD_ARG;
    hipLaunchKernelGGL(sqdist_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, W, U, lengthscale, Wp,
                       dlengthscale, N);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_gram_f32(const float* A, const float* Bm, float* E, int B, int M, int N, int D, int kind,
                            const float* lengthscale, void* stream) {
    if (!A || !E || B <= 0 || M <= 0 || N <= 0 || D <= 0) return DKT_ERR_BAD_ARG;
    // LINEAR_UNIT = LINEAR plus the caller's promise |a_ik| <= 1: it only widens the choice of kernels
    const bool unit = (kind == DKT_KERNEL_LINEAR_UNIT);
    if (unit) kind = DKT_KERNEL_LINEAR;
    if (kind != DKT_KERNEL_LINEAR && kind != DKT_KERNEL_RBF && kind != DKT_KERNEL_SQDIST) return DKT_ERR_BAD_ARG;
    if (kind != DKT_KERNEL_LINEAR && !lengthscale) return DKT_ERR_BAD_ARG;
    const bool sym = (Bm == nullptr);
    if (sym && M != N) return DKT_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    const char* sm = dkt_variant_env("DKT_GRAM_SMALL");
    const bool small_ok = !(sm && sm[0] == '0');
    if (sym && small_ok && dkt_gram_small_launch(A, E, B, N, D, kind, lengthscale, st))
        return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    if (sym && kind != DKT_KERNEL_LINEAR && dkt_gram_dist_ep_launch(A, E, B, N, D, kind, lengthscale, st))
        return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    // a handful of episodes: one workgroup per output tile (the kernels below would leave the chip idle behind one latency chain per episode)
    if (sym && kind == DKT_KERNEL_LINEAR && N > 32 && dkt_gram_fewep_applies(B, N, D) && !((uintptr_t)A & 15)) {
        const int nt = (N + 15) / 16;
        hipLaunchKernelGGL(gram_sym_fewep_kernel, dim3(nt * (nt + 1) / 2, B), dim3(64 * FEW_WAVES), 0, st, A, E, N, D);
        return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    }
    if (sym && kind == DKT_KERNEL_LINEAR && dkt_gram_sym_ep_launch(A, E, B, N, D, unit, st))
        return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    if (sym && kind == DKT_KERNEL_LINEAR && dkt_gram_sym_big_launch(A, E, B, N, D, unit, st))
        return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    if (B > 65535) return DKT_ERR_TOO_LARGE;
    dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, B), block(256);
    if (kind == DKT_KERNEL_LINEAR) {
        if (sym) hipLaunchKernelGGL((gram_nt_kernel<DKT_KERNEL_LINEAR, true>), grid, block, 0, st, A, A, E, M, N, D, lengthscale);
        else hipLaunchKernelGGL((gram_nt_kernel<DKT_KERNEL_LINEAR, false>), grid, block, 0, st, A, Bm, E, M, N, D, lengthscale);
    } else if (kind == DKT_KERNEL_RBF) {
        if (sym) hipLaunchKernelGGL((gram_nt_kernel<DKT_KERNEL_RBF, true>), grid, block, 0, st, A, A, E, M, N, D, lengthscale);
        else hipLaunchKernelGGL((gram_nt_kernel<DKT_KERNEL_RBF, false>), grid, block, 0, st, A, Bm, E, M, N, D, lengthscale);
    } else {
        if (sym) hipLaunchKernelGGL((gram_nt_kernel<DKT_KERNEL_SQDIST, true>), grid, block, 0, st, A, A, E, M, N, D, lengthscale);
        else hipLaunchKernelGGL((gram_nt_kernel<DKT_KERNEL_SQDIST, false>), grid, block, 0, st, A, Bm, E, M, N, D, lengthscale);
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_gram_bwd_f32(const float* W, const float* Z, float* dZ, int B, int N, int D,
                                const float* ep_scale, unsigned flags, void* stream) {
    if (!W || !Z || !dZ || B <= 0 || N <= 0 || D <= 0 || (flags & ~(DKT_GRAM_UNIT_ROWS | DKT_GRAM_W_SYMMETRIC))) return DKT_ERR_BAD_ARG;
    {
        const char* sm = dkt_variant_env("DKT_GRAM_SMALL");
        if (!(sm && sm[0] == '0') && dkt_gram_small_bwd_launch(W, Z, dZ, B, N, D, ep_scale, (hipStream_t)stream))
            return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    }
    if (dkt_gram_bwd_ep_launch(W, Z, dZ, B, N, D, ep_scale, (flags & DKT_GRAM_UNIT_ROWS) != 0, (hipStream_t)stream))
        return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    if (dkt_gram_bwd_big_launch(W, Z, dZ, B, N, D, ep_scale, flags, (hipStream_t)stream))
        return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    if (B > 65535) return DKT_ERR_TOO_LARGE;
    dim3 grid((D + GT - 1) / GT, (N + GT - 1) / GT, B), block(256);
    hipLaunchKernelGGL(gram_bwd_kernel, grid, block, 0, (hipStream_t)stream, W, Z, dZ, N, D, ep_scale);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_rbf_bwd_f32(const float* W, const float* E, const float* lengthscale, float* Wp,
                               float* dlengthscale, int B, int N, void* stream) {
    if (!W || !E || !lengthscale || !Wp || !dlengthscale || B <= 0 || N <= 0) return DKT_ERR_BAD_ARG;
    hipLaunchKernelGGL(rbf_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, W, E, lengthscale, Wp,
                       dlengthscale, N);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
