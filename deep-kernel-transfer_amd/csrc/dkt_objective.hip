// dkt_objective.hip -- the small reductions of a training step (gfx950; the last section: the bn_out parameter gradients of the fused front end).  First the two
// [B, C]-sized reductions around dkt_mll_f32: the episode's objective from the class models' log marginal likelihoods and the
// chain rule from the per-episode, per-class hyper-parameter gradients to the [C] parameters.  Arithmetically nothing -- but as tensor expressions they are seven
// launches (two multiplies + a reduction in the forward, a multiply and two multiply + reduction pairs in the backward), a third of the launches of a training step,
// and the reference's literal loop (one episode per step, methods/DKT.py:117-164) is bound by exactly that.  Fixed summation order: bitwise reproducible.
#include "dkt_common.h"
#include "../../include/dkt_abi.h"

namespace {

__global__ __launch_bounds__(256) void objective_kernel(const float* __restrict__ logp, const float* __restrict__ cw, float* __restrict__ obj, const int B, const int C) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += (cw ? cw[c] : 1.0f) * logp[(size_t)b * C + c];
    obj[b] = s;
}

// grid = (C, 3): workgroup (c, x) folds column c of d_x[B, C] against gobj[B]; a thread sums its episodes b = t, t + 256, ... in order, the threads meet in a fixed tree
__global__ __launch_bounds__(256) void hyper_grads_kernel(const float* __restrict__ gobj, const float* __restrict__ cw, const float* __restrict__ d0, const float* __restrict__ d1,
                                                          const float* __restrict__ d2, float* __restrict__ g0, float* __restrict__ g1, float* __restrict__ g2, const int B,
                                                          const int C) {
    __shared__ float red[256];
    const int c = blockIdx.x, which = blockIdx.y, tid = threadIdx.x;
    const float* d = which == 0 ? d0 : (which == 1 ? d1 : d2);
    float* g = which == 0 ? g0 : (which == 1 ? g1 : g2);
    if (!d || !g) return;                                        // uniform
    float s = 0.f;
    for (int b = tid; b < B; b += 256) s += gobj[b] * d[(size_t)b * C + c];
    red[tid] = s;
    __syncthreads();
#pragma unroll
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) g[c] = (cw ? cw[c] : 1.0f) * red[0];
}

// Column sums of the per-episode parts of the bn_out parameter gradients: part0 / part1 [B, D] -> out0 / out1 [D] (or, with more than one row chunk, the chunk's
// partial sums [R, 2, D]).  grid = (ceil(D / 64), R): workgroup (x, r) = 16 feature quads x 64 row lanes; a row lane sums its rows r 256 + rl, + 64, ... in order, the
// lanes meet in a fixed tree -- bitwise reproducible.
__global__ __launch_bounds__(1024) void bn_param_grads_kernel(const float* __restrict__ p0, const float* __restrict__ p1, float* __restrict__ o0, float* __restrict__ o1,
                                                              const int B, const int D, const long o_rstride) {
    __shared__ __attribute__((aligned(16))) float red[64][16][8];
    const int qd = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int d = 64 * blockIdx.x + 4 * qd;
    const int r0 = 256 * blockIdx.y, r1 = min(B, r0 + 256);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (d < D) {                                               // (D % 4 == 0: a quad is inside the row or wholly beyond it)
#pragma unroll 4
        for (int b = r0 + rl; b < r1; b += 64) {
            const float4 u = *reinterpret_cast<const float4*>(p0 + (size_t)b * D + d), v = *reinterpret_cast<const float4*>(p1 + (size_t)b * D + d);
            s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
            s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        }
    }
    *reinterpret_cast<float4*>(&red[rl][qd][0]) = s0;
    *reinterpret_cast<float4*>(&red[rl][qd][4]) = s1;
    __syncthreads();
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) {
        if (rl < w) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[rl][qd][e] += red[rl + w][qd][e];
        }
        __syncthreads();
    }
    if (rl == 0 && d < D) {
        *reinterpret_cast<float4*>(o0 + (size_t)blockIdx.y * o_rstride + d) = *reinterpret_cast<const float4*>(&red[0][qd][0]);
        *reinterpret_cast<float4*>(o1 + (size_t)blockIdx.y * o_rstride + d) = *reinterpret_cast<const float4*>(&red[0][qd][4]);
    }
}

// the R row chunks' partial sums ws[R][2][D] -> out0 / out1 [D], chunk by chunk in order
__global__ __launch_bounds__(256) void bn_param_grads_fold_kernel(const float* __restrict__ ws, float* __restrict__ o0, float* __restrict__ o1, const int R, const int D) {
    const int d = 4 * (blockIdx.x * 256 + threadIdx.x);
    if (d >= D) return;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    for (int r = 0; r < R; ++r) {
        const float4 u = *reinterpret_cast<const float4*>(ws + ((size_t)2 * r) * D + d), v = *reinterpret_cast<const float4*>(ws + ((size_t)2 * r + 1) * D + d);
        s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
    }
    *reinterpret_cast<float4*>(o0 + d) = s0;
    *reinterpret_cast<float4*>(o1 + d) = s1;
}

}  // namespace

extern "C" size_t dkt_bn_param_grads_workspace_bytes(int B, int D) {
    if (B <= 256 || D <= 0) return 0;
    return (size_t)((B + 255) / 256) * 2 * (size_t)D * sizeof(float);
}

extern "C" int dkt_bn_param_grads_f32(const float* dgamma_part, const float* dbeta_part, float* dgamma, float* dbeta, int B, int D, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    if (!dgamma_part || !dbeta_part || !dgamma || !dbeta || B <= 0 || D <= 0 || (D & 3)) return DKT_ERR_BAD_ARG;
    if (((uintptr_t)dgamma_part & 15) || ((uintptr_t)dbeta_part & 15) || ((uintptr_t)dgamma & 15) || ((uintptr_t)dbeta & 15) || ((uintptr_t)workspace & 15)) return DKT_ERR_BAD_ARG;
    const int R = (B + 255) / 256;
    if (R > 65535) return DKT_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((D + 63) / 64, R);
    if (R == 1) {
        hipLaunchKernelGGL(bn_param_grads_kernel, grid, dim3(1024), 0, st, dgamma_part, dbeta_part, dgamma, dbeta, B, D, 0L);
    } else {
        if (!workspace || workspace_bytes < dkt_bn_param_grads_workspace_bytes(B, D)) return DKT_ERR_WORKSPACE;
        float* ws = (float*)workspace;
        hipLaunchKernelGGL(bn_param_grads_kernel, grid, dim3(1024), 0, st, dgamma_part, dbeta_part, ws, ws + D, B, D, 2L * D);
        hipLaunchKernelGGL(bn_param_grads_fold_kernel, dim3((D / 4 + 255) / 256), dim3(256), 0, st, ws, dgamma, dbeta, R, D);
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_objective_f32(const float* logp, const float* cls_weight, float* obj, int B, int C, void* stream) {
    if (!logp || !obj || B <= 0 || C <= 0) return DKT_ERR_BAD_ARG;
    hipLaunchKernelGGL(objective_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, logp, cls_weight, obj, B, C);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_hyper_grads_f32(const float* gobj, const float* cls_weight, const float* dsv, const float* dmean, const float* dnoise, float* gsv, float* gmean,
                                   float* gnoise, int B, int C, void* stream) {
    if (!gobj || B <= 0 || C <= 0 || C > 65535 || (!dsv != !gsv) || (!dmean != !gmean) || (!dnoise != !gnoise)) return DKT_ERR_BAD_ARG;
    if (!gsv && !gmean && !gnoise) return DKT_OK;
    hipLaunchKernelGGL(hyper_grads_kernel, dim3(C, 3), dim3(256), 0, (hipStream_t)stream, gobj, cls_weight, dsv, dmean, dnoise, gsv, gmean, gnoise, B, C);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
