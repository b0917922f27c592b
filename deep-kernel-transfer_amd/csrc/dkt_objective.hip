// dkt_objective.hip -- the two [B, C]-sized reductions around dkt_mll_f32 (gfx950): the episode's objective from the class models' log marginal likelihoods and the
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

}  // namespace

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
