// Spectral-mixture base kernel of the regression head (reference methods/DKT_regression.py:121-122:
// gpytorch.kernels.SpectralMixtureKernel(num_mixtures=4, ard_num_dims=2916), no ScaleKernel around it):
//
//     E[i,j] = sum_q w_q  prod_d exp(-2 pi^2 (sigma_qd tau_d)^2) cos(2 pi mu_qd tau_d),     tau = a_i - b_j
//
// and its chain rule towards the features, the mixture weights, means and scales.  gfx950 only.
//
// The product over D = 2916 cosines of magnitude < 1 leaves the fp32 range for all but near-identical inputs, so a mixture
// term is carried as (sign, log magnitude): E_q = sgn * exp(-2 pi^2 sum_d (sigma tau)^2 + sum_d log|cos|).  Terms that
// underflow anyway (GPyTorch's plain fp32 product returns 0 for them) come out as 0 here too; the ones in range keep a
// relative accuracy of ~1e-6 instead of the D-fold rounding of a running product.  cos(2 pi u) is evaluated as
// cospi(2u): the argument reduction is exact.
#include "dkt_common.h"
#include "../../include/dkt_abi.h"

#define SMK_MAXQ 8
#define SMK_2PI2 19.739208802178716f   // 2 pi^2
#define SMK_4PI2 39.478417604357432f   // 4 pi^2
#define SMK_2PI 6.2831853071795865f

// one workgroup per (i, j) entry; threads stride over the D features, fixed-order reductions (bitwise reproducible)
template <int Q>
__global__ __launch_bounds__(256) void smk_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                      const float* __restrict__ wgt, const float* __restrict__ mu,
                                                      const float* __restrict__ sg, float* __restrict__ E,
                                                      float* __restrict__ Eq, int M, int N, int D, int sym) {
    const int j = blockIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (sym && j > i) return;
    __shared__ float red[4][2 * Q];
    __shared__ unsigned redn[4];
    const float* a = x1 + ((size_t)b * M + i) * D;
    const float* c = x2 + ((size_t)b * N + j) * D;
    float S[Q], L[Q];
    unsigned neg = 0;
#pragma unroll
    for (int q = 0; q < Q; ++q) S[q] = 0.f, L[q] = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float tau = a[d] - c[d];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float s = sg[(size_t)q * D + d] * tau;
            S[q] = __builtin_fmaf(s, s, S[q]);
            const float cv = cospif(2.f * mu[(size_t)q * D + d] * tau);
            L[q] += logf(fabsf(cv));
            neg ^= (cv < 0.f ? 1u : 0u) << q;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) neg ^= (unsigned)__shfl_xor((int)neg, o, DKT_WAVE);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const float s = wave_sum(S[q]), l = wave_sum(L[q]);
        if (lane == 0) red[wave][2 * q] = s, red[wave][2 * q + 1] = l;
    }
    if (lane == 0) redn[wave] = neg;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned n = redn[0] ^ redn[1] ^ redn[2] ^ redn[3];
        float e = 0.f;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const float s = (red[0][2 * q] + red[1][2 * q]) + (red[2][2 * q] + red[3][2 * q]);
            const float l = (red[0][2 * q + 1] + red[1][2 * q + 1]) + (red[2][2 * q + 1] + red[3][2 * q + 1]);
            float eq = expf(-SMK_2PI2 * s + l);
            if ((n >> q) & 1u) eq = -eq;
            if (Eq) {
                Eq[(((size_t)b * Q + q) * M + i) * N + j] = eq;
                if (sym && i != j) Eq[(((size_t)b * Q + q) * M + j) * N + i] = eq;
            }
            e = __builtin_fmaf(wgt[q], eq, e);
        }
        E[((size_t)b * M + i) * N + j] = e;
        if (sym && i != j) E[((size_t)b * M + j) * N + i] = e;
    }
}

// Chain rule for the symmetric case (x1 == x2 == x): a thread owns one feature d, a wave one residue class of the row
// index i (so every coefficient load is wave-uniform); the 64-feature slab of x sits in LDS.
//   dE_q/dtau_d   = E_q (-4 pi^2 sigma^2 tau - 2 pi mu tan(2 pi mu tau))
//   dE_q/dmu_d    = E_q (-2 pi tau tan(2 pi mu tau))
//   dE_q/dsigma_d = E_q (-4 pi^2 sigma tau^2)
template <int Q>
__global__ __launch_bounds__(256) void smk_bwd_kernel(const float* __restrict__ gE, const float* __restrict__ Eq,
                                                      const float* __restrict__ x, const float* __restrict__ wgt,
                                                      const float* __restrict__ mu, const float* __restrict__ sg,
                                                      float* __restrict__ dx, float* __restrict__ dmu,
                                                      float* __restrict__ dsg, int N, int D) {
    extern __shared__ float xs[];                       // [N][64] slab, then [3][2Q][64] for the cross-wave sums
    const int b = blockIdx.y, dl = threadIdx.x & 63, ig = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl;
    const bool ok = d < D;
    const float* xb = x + (size_t)b * N * D;
    for (int i = ig; i < N; i += 4) xs[i * 64 + dl] = ok ? xb[(size_t)i * D + d] : 0.f;
    float m[Q], s[Q], am[Q], as[Q], w[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        m[q] = ok ? mu[(size_t)q * D + d] : 0.f;
        s[q] = ok ? sg[(size_t)q * D + d] : 0.f;
        w[q] = wgt[q];
        am[q] = 0.f, as[q] = 0.f;
    }
    __syncthreads();
    const float* g = gE + (size_t)b * N * N;
    const float* eqb = Eq + (size_t)b * Q * N * N;
    for (int i = ig; i < N; i += 4) {
        const float xi = xs[i * 64 + dl];
        float acc = 0.f;
        for (int j = 0; j < N; ++j) {
            if (j == i) continue;
            const float g1 = g[(size_t)i * N + j], g2 = g[(size_t)j * N + i];
            const float tau = xi - xs[j * 64 + dl];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float cw = w[q] * eqb[((size_t)q * N + i) * N + j];
                if (cw == 0.f) continue;                // wave-uniform: the term underflowed
                float sn, cs;
                sincospif(2.f * m[q] * tau, &sn, &cs);
                if (fabsf(cs) < 1e-30f) cs = copysignf(1e-30f, cs);
                const float t = sn / cs;
                const float st = s[q] * tau;
                am[q] = __builtin_fmaf(g1 * cw, -SMK_2PI * tau * t, am[q]);
                as[q] = __builtin_fmaf(g1 * cw, -SMK_4PI2 * st * tau, as[q]);
                acc = __builtin_fmaf((g1 + g2) * cw, -SMK_4PI2 * s[q] * st - SMK_2PI * m[q] * t, acc);
            }
        }
        if (ok) dx[((size_t)b * N + i) * D + d] = acc;
    }
    float* part = xs + (size_t)N * 64;
    if (ig > 0) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            part[((ig - 1) * 2 * Q + 2 * q) * 64 + dl] = am[q];
            part[((ig - 1) * 2 * Q + 2 * q + 1) * 64 + dl] = as[q];
        }
    }
    __syncthreads();
    if (ig == 0 && ok) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float a0 = am[q], a1 = as[q];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                a0 += part[(k * 2 * Q + 2 * q) * 64 + dl];
                a1 += part[(k * 2 * Q + 2 * q + 1) * 64 + dl];
            }
            dmu[((size_t)b * Q + q) * D + d] = a0;
            dsg[((size_t)b * Q + q) * D + d] = a1;
        }
    }
}

template <int Q>
static int smk_fwd_launch(const float* x1, const float* x2, const float* w, const float* mu, const float* sg, float* E,
                          float* Eq, int B, int M, int N, int D, int sym, hipStream_t st) {
    hipLaunchKernelGGL((smk_fwd_kernel<Q>), dim3(N, M, B), dim3(256), 0, st, x1, x2, w, mu, sg, E, Eq, M, N, D, sym);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

template <int Q>
static int smk_bwd_launch(const float* gE, const float* Eq, const float* x, const float* w, const float* mu,
                          const float* sg, float* dx, float* dmu, float* dsg, int B, int N, int D, hipStream_t st) {
    const size_t lds = ((size_t)N * 64 + 3 * 2 * Q * 64) * sizeof(float);
    if (lds > 64 * 1024) return DKT_ERR_TOO_LARGE;
    hipLaunchKernelGGL((smk_bwd_kernel<Q>), dim3((D + 63) / 64, B), dim3(256), lds, st, gE, Eq, x, w, mu, sg, dx, dmu,
                       dsg, N, D);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

#define SMK_DISPATCH(Qv, CALL)                          \
    switch (Qv) {                                       \
        case 1: return CALL(1);                         \
        case 2: return CALL(2);                         \
        case 3: return CALL(3);                         \
        case 4: return CALL(4);                         \
        case 5: return CALL(5);                         \
        case 6: return CALL(6);                         \
        case 7: return CALL(7);                         \
        case 8: return CALL(8);                         \
        default: return DKT_ERR_TOO_LARGE;              \
    }

extern "C" int dkt_smk_f32(const float* x1, const float* x2, const float* weights, const float* means,
                           const float* scales, float* E, float* Eq, int B, int M, int N, int D, int Q, void* stream) {
    if (!x1 || !weights || !means || !scales || !E || B <= 0 || M <= 0 || N <= 0 || D <= 0 || Q <= 0)
        return DKT_ERR_BAD_ARG;
    if (!x2 && M != N) return DKT_ERR_BAD_ARG;
    if (B > 65535 || M > 65535) return DKT_ERR_TOO_LARGE;
    const int sym = x2 ? 0 : 1;
    const float* xb = x2 ? x2 : x1;
    hipStream_t st = (hipStream_t)stream;
#define SMK_F(QQ) smk_fwd_launch<QQ>(x1, xb, weights, means, scales, E, Eq, B, M, N, D, sym, st)
    SMK_DISPATCH(Q, SMK_F)
#undef SMK_F
}

extern "C" int dkt_smk_bwd_f32(const float* gE, const float* Eq, const float* x, const float* weights,
                               const float* means, const float* scales, float* dx, float* dmeans, float* dscales,
                               int B, int N, int D, int Q, void* stream) {
    if (!gE || !Eq || !x || !weights || !means || !scales || !dx || !dmeans || !dscales || B <= 0 || N <= 0 ||
        D <= 0 || Q <= 0)
        return DKT_ERR_BAD_ARG;
    if (B > 65535) return DKT_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
#define SMK_B(QQ) smk_bwd_launch<QQ>(gE, Eq, x, weights, means, scales, dx, dmeans, dscales, B, N, D, st)
    SMK_DISPATCH(Q, SMK_B)
#undef SMK_B
}
