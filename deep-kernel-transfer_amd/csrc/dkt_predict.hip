// dkt_predict.hip -- eval-mode exact-GP prediction: posterior mean of the C one-vs-rest models at
// M test points + arg-max label, and the predictive variance diagonal.
//
// Replaces `self.likelihood(*self.model(*z_query_list))` -> gaussian.mean -> sigmoid -> vstack ->
// argmax (reference methods/DKT.py:176-181, 186-191, 264-270, 329-334) and
// `pred = self.likelihood(self.model(z_query)); pred.confidence_region()`
// (methods/DKT_regression.py:92-93), i.e. GPyTorch DefaultPredictionStrategy.exact_prediction.
#include "dkt_common.h"
#include "../../include/dkt_abi.h"

namespace {

// grid = (ceil(M / 256), B); dynamic LDS = C*N floats (the episode's mean caches).
__global__ __launch_bounds__(256) void predict_mean_kernel(const float* __restrict__ Ex, const float* __restrict__ alpha,
                                                           const float* __restrict__ sv, const float* __restrict__ mean,
                                                           float* __restrict__ mu, int32_t* __restrict__ labels,
                                                           int C, int M, int N, long ex_cstride) {
    extern __shared__ __attribute__((aligned(16))) float s_alpha[];
    const int b = blockIdx.y;
    const float* ab = alpha + (size_t)b * C * N;
    for (int i = threadIdx.x; i < C * N; i += blockDim.x) s_alpha[i] = ab[i];
    __syncthreads();
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= M) return;
    // ex_cstride = 0: one base cross kernel shared by the class models (Ex [B,M,N]); M N: one per class model (Ex [B,C,M,N])
    const float* ex0 = Ex + ((size_t)b * (ex_cstride ? C : 1) * M + qi) * N;
    float best = 0.f;
    int best_c = 0;
    for (int c = 0; c < C; ++c) {
        const float* ac = s_alpha + c * N;
        const float* ex = ex0 + (size_t)c * ex_cstride;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += ex[n] * ac[n];
        const float v = mean[c] + sv[c] * s;
        mu[((size_t)b * C + c) * M + qi] = v;
        if (c == 0 || v > best) { best = v; best_c = c; }   // first maximum wins (np.argmax)
    }
    if (labels) labels[(size_t)b * M + qi] = best_c;
}

// var[b,c,q] = sv_c exx[b,q] - sv_c^2 |L_c^-1 ex_q|^2 + noise_c.  grid = (ceil(M/64), C, B), block 64;
// each thread forward-substitutes its own right-hand side, kept in LDS (64 * N floats).
__global__ __launch_bounds__(64) void predict_var_kernel(const float* __restrict__ Ex, const float* __restrict__ exx,
                                                         const float* __restrict__ L, const float* __restrict__ sv,
                                                         const float* __restrict__ noise, float* __restrict__ var,
                                                         int C, int M, int N) {
    extern __shared__ __attribute__((aligned(16))) float s_v[];   // [N][64]
    const int b = blockIdx.z, c = blockIdx.y;
    const int qi = blockIdx.x * 64 + threadIdx.x;
    if (qi >= M) return;
    const float* ex = Ex + ((size_t)b * M + qi) * N;
    const float* Lc = L + ((size_t)b * C + c) * N * N;
    const float s = sv[c];
    float acc2 = 0.f;
    for (int i = 0; i < N; ++i) {
        float r = s * ex[i];
        for (int k = 0; k < i; ++k) r -= Lc[(size_t)i * N + k] * s_v[k * 64 + threadIdx.x];
        r /= Lc[(size_t)i * N + i];
        s_v[i * 64 + threadIdx.x] = r;
        acc2 += r * r;
    }
    var[((size_t)b * C + c) * M + qi] = s * exx[(size_t)b * M + qi] - acc2 + noise[c];
}

}  // namespace

static int predict_mean_launch(const float* Ex, const float* alpha, const float* sv, const float* mean,
                               float* mu, int32_t* labels, int B, int C, int M, int N, long ex_cstride, void* stream) {
    if (!Ex || !alpha || !sv || !mean || !mu || B <= 0 || C <= 0 || M <= 0 || N <= 0) return DKT_ERR_BAD_ARG;
    const size_t lds = (size_t)C * N * sizeof(float);
    if (lds > 150 * 1024 || B > 65535) return DKT_ERR_TOO_LARGE;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void*)predict_mean_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DKT_ERR_LAUNCH;
    hipLaunchKernelGGL(predict_mean_kernel, dim3((M + 255) / 256, B), dim3(256), lds, (hipStream_t)stream, Ex,
                       alpha, sv, mean, mu, labels, C, M, N, ex_cstride);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_predict_f32(const float* Ex, const float* alpha, const float* sv, const float* mean,
                               float* mu, int32_t* labels, int B, int C, int M, int N, void* stream) {
    return predict_mean_launch(Ex, alpha, sv, mean, mu, labels, B, C, M, N, 0, stream);
}

extern "C" int dkt_predict_per_class_f32(const float* Ex, const float* alpha, const float* sv, const float* mean,
                                         float* mu, int32_t* labels, int B, int C, int M, int N, void* stream) {
    return predict_mean_launch(Ex, alpha, sv, mean, mu, labels, B, C, M, N, (long)M * N, stream);
}

extern "C" int dkt_predict_var_f32(const float* Ex, const float* exx, const float* L, const float* sv,
                                   const float* noise, float* var, int B, int C, int M, int N, void* stream) {
    if (!Ex || !exx || !L || !sv || !noise || !var || B <= 0 || C <= 0 || M <= 0 || N <= 0) return DKT_ERR_BAD_ARG;
    const size_t lds = (size_t)N * 64 * sizeof(float);
    if (lds > 150 * 1024 || B > 65535 || C > 65535) return DKT_ERR_TOO_LARGE;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void*)predict_var_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DKT_ERR_LAUNCH;
    hipLaunchKernelGGL(predict_var_kernel, dim3((M + 63) / 64, C, B), dim3(64), lds, (hipStream_t)stream, Ex, exx,
                       L, sv, noise, var, C, M, N);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_abi_version(void) { return DKT_ABI_VERSION; }

extern "C" int dkt_device_cu_count(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    return prop.multiProcessorCount;
}
