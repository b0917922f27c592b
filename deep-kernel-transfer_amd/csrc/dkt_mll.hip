// dkt_mll.hip -- exact-GP marginal log likelihood of the C one-vs-rest models of an episode:
// noise add, jittered Cholesky, log-det, mean cache alpha = K^-1 (y - m), and (training) the
// gradient pieces W = d obj / d E, d logp / d(sv, mean, noise).
//
// GENERIC path (any N): one 256-thread workgroup per episode, classes processed in sequence
// (the episode's W is accumulated by the same threads class after class -- no atomics, bitwise
// reproducible).  The (N+1) x N working matrix lives in LDS when it fits (N <= ~190) and in a
// caller-provided global workspace otherwise (N = 320/420 stress shapes).
//
// Working matrix Mw (row-major, leading dimension LD odd):
//   rows 0..N-1 : lower triangle (p >= j) = K_c, overwritten by L;  strict upper triangle (p < j)
//                 = the rows of an appended identity, which the same column sweep turns into
//                 U = L^-T (so K^-1 = U U^T and alpha = U w need no separate triangular solve);
//   row  N      : r = y_c - m_c, turned into w = L^-1 r by the sweep.
// Right-looking sweep, column k:  d = Mw[k][k] (fail if !(d > 0), as LAPACK potrf / torch.cholesky);
//   Mw[:,k] *= 1/sqrt(d);  Mw[p][j] -= Mw[p][k] Mw[j][k]  for j > k and (p >= j or p <= k).
//
// Replaces `-self.mll(output, targets)` + backward + the eval-mode mean cache
// (reference methods/DKT.py:161-163, 177, 187, 252-254, 265, 330; methods/DKT_regression.py:53-56, 92),
// i.e. GPyTorch psd_safe_cholesky / inv_quad_logdet / cholesky_solve.
#include "dkt_mll.h"
#include <cstdlib>

namespace {

constexpr size_t MLL_LDS_LIMIT = 150 * 1024;

__host__ __device__ inline int mll_ld(int N) { return N | 1; }
inline size_t mll_vec_floats(int N) { return (size_t)3 * (N + 1) + 32; }
inline size_t mll_mat_floats(int N) { return (size_t)(N + 1) * mll_ld(N); }
inline bool mll_fits_lds(int N) { return (mll_vec_floats(N) + mll_mat_floats(N)) * 4 <= MLL_LDS_LIMIT; }

template <bool GLOBAL>
__global__ __launch_bounds__(256) void mll_generic_kernel(MllArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int N = a.N, C = a.C, LD = a.LD, R = N + 1;
    // DKT_MLL_E_PER_CLASS: a workgroup is ONE (episode, class) matrix -- its own E[b, c], its own W[b, c], a class "loop" of one entry
    // (workgroup x = matrix b0 C + x); otherwise a workgroup is an episode (b0 + x) and walks its C classes
    const bool epc = (a.flags & DKT_MLL_E_PER_CLASS) != 0;
    const size_t unit = epc ? (size_t)a.b0 * C + blockIdx.x : (size_t)a.b0 + blockIdx.x;
    const int b = epc ? (int)(unit / (size_t)C) : (int)unit;
    const int c_first = epc ? (int)(unit % (size_t)C) : 0, c_end = epc ? c_first + 1 : C;
    if (a.only_failed) {                       // fix-up pass of the blocked / tile-array / band / f16-split paths: nothing to do for a unit without a flag
        bool need = false;
        for (int c = c_first; c < c_end; ++c) need = need || a.only_failed[(size_t)b * C + c] != 0;
        if (!need) return;                     // (uniform)
        __syncthreads();                       // every thread has read the flags before anybody rewrites info[]
    }
    float* cs = smem;            // [R]  scaled column of the current sweep step
    float* ldiag = cs + R;       // [R]  diag(L)
    float* al = ldiag + R;       // [R]  alpha
    float* red = al + R;         // [32] reduction scratch
    float* Mw = GLOBAL ? (a.ws + (size_t)blockIdx.x * R * LD) : (red + 32);
#define MW(p, j) Mw[(p) * LD + (j)]

    const float* Eb = a.E + unit * N * N;
    const bool want_grad = (a.flags & DKT_MLL_WANT_GRAD) != 0;
    const bool want_chol = (a.flags & DKT_MLL_WANT_CHOL) != 0;
    float* Wb = want_grad ? a.W + unit * N * N : nullptr;

    for (int c = c_first; c < c_end; ++c) {
        const float svc = a.sv[c], mc = a.mean[c], nzc = a.noise[c];
        const float* yc = a.Y + (size_t)b * a.y_bstride + (size_t)c * N;
        int fail_at = 0;
        float jit = 0.f;
        for (int attempt = 0; attempt <= a.max_tries; ++attempt) {
            jit = 0.f;
            if (attempt > 0) {
                jit = a.jitter0;
                for (int i = 1; i < attempt; ++i) jit *= 10.f;
            }
            __syncthreads();
            // ---- form the working matrix ----
            for (int idx = tid; idx < N * N; idx += 256) {
                const int p = idx / N, j = idx - p * N;
                float v = 0.f;
                if (p >= j) {
                    v = svc * Eb[idx];
                    if (p == j) v += nzc + jit;
                }
                MW(p, j) = v;
            }
            for (int j = tid; j < N; j += 256) MW(N, j) = yc[j] - mc;
            __syncthreads();
            // ---- right-looking sweep ----
            fail_at = 0;
            for (int k = 0; k < N; ++k) {
                const float d = MW(k, k);
                if (!(d > 0.f)) { fail_at = k + 1; break; }   // block-uniform (same LDS/global word)
                const float lkk = sqrtf(d);
                const float rinv = 1.0f / lkk;
                // column k (all rows but the pivot itself, which stays untouched until after the sweep,
                // so the pivot read above needs no extra barrier)
                for (int p = tid; p < R; p += 256) {
                    if (p != k) {
                        const float v = MW(p, k) * rinv;
                        cs[p] = v;
                        MW(p, k) = v;
                    } else {
                        cs[p] = rinv;
                        ldiag[k] = lkk;
                    }
                }
                __syncthreads();
                for (int p = ty; p < R; p += 16) {
                    const bool low = p <= k;
                    const float cp = cs[p];
                    for (int j = k + 1 + tx; j < N; j += 16) {
                        if (low || p >= j) MW(p, j) -= cp * cs[j];
                    }
                }
                __syncthreads();
            }
            if (fail_at == 0) break;
        }
        __syncthreads();
        const size_t bc = (size_t)b * C + c;
        if (fail_at != 0) {
            // not positive definite after every jitter retry: poison the outputs (GPyTorch raises NotPSDError)
            const float qnan = __int_as_float(0x7fc00000);
            if (tid == 0) {
                a.logp[bc] = qnan;
                a.jitter_used[bc] = jit;
                a.info[bc] = fail_at;
                if (want_grad) { a.dsv[bc] = qnan; a.dmean[bc] = qnan; a.dnoise[bc] = qnan; }
            }
            for (int i = tid; i < N; i += 256) a.alpha[bc * N + i] = qnan;
            if (want_grad)
                for (int idx = tid; idx < N * N; idx += 256) Wb[idx] = qnan;
            if (want_chol)
                for (int idx = tid; idx < N * N; idx += 256) a.L[bc * N * N + idx] = qnan;
            continue;
        }
        for (int k = tid; k < N; k += 256) MW(k, k) = 1.0f / ldiag[k];   // U_kk = 1 / L_kk
        __syncthreads();
        // ---- quad form, log-det, alpha = U w ----
        float qpart = 0.f, lpart = 0.f;
        for (int k = tid; k < N; k += 256) {
            const float w = MW(N, k);
            qpart += w * w;
            lpart += logf(ldiag[k]);
        }
        const float quad = block_sum_256(qpart, red);
        const float logdet_half = block_sum_256(lpart, red);
        float apart = 0.f;
        for (int i = tid; i < N; i += 256) {
            float s = 0.f;
            for (int m = i; m < N; ++m) s += MW(i, m) * MW(N, m);
            al[i] = s;
            a.alpha[bc * N + i] = s;
            apart += s;
        }
        const float asum = block_sum_256(apart, red);   // also orders al[] before the readers below
        if (tid == 0) {
            a.logp[bc] = -0.5f * quad - logdet_half - (float)N * DKT_HALF_LOG_2PI;
            a.jitter_used[bc] = jit;
            a.info[bc] = 0;
        }
        if (want_chol) {
            float* Lb = a.L + bc * N * N;
            for (int idx = tid; idx < N * N; idx += 256) {
                const int p = idx / N, j = idx - p * N;
                Lb[idx] = (p > j) ? MW(p, j) : (p == j ? ldiag[p] : 0.f);
            }
        }
        if (want_grad) {
            const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
            const float coef = cw * svc;
            float dsv_part = 0.f, tr_part = 0.f;
            for (int i = ty; i < N; i += 16) {
                for (int j = tx; j <= i; j += 16) {
                    float kinv = 0.f;
                    for (int m = i; m < N; ++m) kinv += MW(i, m) * MW(j, m);
                    const float mcv = 0.5f * (al[i] * al[j] - kinv);
                    const float e = Eb[(size_t)i * N + j];
                    if (i == j) { dsv_part += mcv * e; tr_part += mcv; }
                    else dsv_part += 2.0f * mcv * e;
                    const float wv = coef * mcv;
                    if (c == c_first) {
                        Wb[(size_t)i * N + j] = wv;
                        if (i != j) Wb[(size_t)j * N + i] = wv;
                    } else {
                        Wb[(size_t)i * N + j] += wv;
                        if (i != j) Wb[(size_t)j * N + i] += wv;
                    }
                }
            }
            const float dsv = block_sum_256(dsv_part, red);
            const float trm = block_sum_256(tr_part, red);
            if (tid == 0) {
                a.dsv[bc] = dsv;
                a.dmean[bc] = asum;
                a.dnoise[bc] = trm;
            }
        }
    }
#undef MW
}

}  // namespace

size_t dkt_mll_generic_global_floats(int count, int N) { return (size_t)count * mll_mat_floats(N); }

void dkt_mll_generic_global_launch(MllArgs a, int b0, int count, float* ws, hipStream_t st) {
    a.b0 = b0; a.ws = ws; a.LD = mll_ld(a.N);
    const int units = (a.flags & DKT_MLL_E_PER_CLASS) ? count * a.C : count;       // per-class base matrices: one workgroup (and one working matrix) per (episode, class)
    hipLaunchKernelGGL(mll_generic_kernel<true>, dim3(units), dim3(256), mll_vec_floats(a.N) * sizeof(float), st, a);
}

// kappa-aware dispatch of the f16-split kernels (N + 1 <= 128): their 22-bit tile products hold the tolerances up to cond(K) of a few
// thousand (the reference's unit-norm rows and frozen noise 0.1: <= 730) and 2 - 3 x the tolerances at 1.6e4 (test_mll_large_and_small_magnitude_base_matrices).  After
// the split launch the generic kernel -- exact fp32 on the matrix itself, jitter ladder included, LDS-resident at these sizes -- redoes every unit (episode, or matrix
// with DKT_MLL_E_PER_CLASS) whose a-priori bound 1 + sv trace(E) / noise exceeds MLL_H2_KAPPA_MAX, decided on the device: the split kernels take the trace off the
// diagonal tiles they hold anyway and leave such a class with info = -1 (MllArgs::kappa_max; a launch of its own until ABI 7), then the generic kernel whose workgroups
// read their flags and leave.  DKT_MLL_NO_KAPPA_GUARD = the raw split kernels (tests, A/B tools).
constexpr float MLL_H2_KAPPA_MAX = 5.0e3f;

static bool mll_kappa_guarded(const MllArgs& a) { return !(a.flags & DKT_MLL_NO_KAPPA_GUARD) && mll_fits_lds(a.N); }

static void mll_kappa_fixup(MllArgs a, hipStream_t st) {
    if (!mll_kappa_guarded(a)) return;
    const int upe = (a.flags & DKT_MLL_E_PER_CLASS) ? a.C : 1;
    const size_t units = (size_t)a.B * upe;
    a.only_failed = a.info;                                    // the generic kernel leaves at once (one flag read per class, no barrier) unless the unit is flagged
    a.b0 = 0; a.LD = mll_ld(a.N);
    const size_t lds = (mll_vec_floats(a.N) + mll_mat_floats(a.N)) * sizeof(float);
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)mll_generic_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return;
    hipLaunchKernelGGL(mll_generic_kernel<false>, dim3((unsigned)units), dim3(256), lds, st, a);
}

static int g_mll_env_read = 0, g_p2_guard = 1, g_force_f32mfma = 0;
void dkt_mll_reload_env() { g_mll_env_read = 0; }         // dkt_reload_env()

extern "C" size_t dkt_mll_workspace_bytes(int B, int C, int N) {
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    if (N + 1 <= 128) return 0;                              // register-resident kernel
    const size_t big = dkt_mll_big_workspace_bytes(B, C, N); // blocked path; also covers the generic kernel's global matrices
    const size_t gen = mll_fits_lds(N) ? 0 : (size_t)B * mll_mat_floats(N) * sizeof(float);
    const size_t til = dkt_mll_tiled_supports(N, 0) ? dkt_mll_tiled_workspace_bytes(B, C, N) : 0;
    const size_t bnd = dkt_mll_band_supports(N, 0, C) ? dkt_mll_band_workspace_bytes(B, C, N) : 0;
    size_t mx = big > gen ? big : gen;
    mx = mx > bnd ? mx : bnd;
    return mx > til ? mx : til;
}

// What THIS call's flags need (the flag-less query reserves the larger of the shared-matrix and per-class tile layouts and the blocked path's
// 4 N^2 floats per matrix for every call; a default shared-matrix call of a few episodes needs a tenth of that).
extern "C" size_t dkt_mll_workspace_bytes_for(int B, int C, int N, unsigned flags) {
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    const bool pc = (flags & DKT_MLL_E_PER_CLASS) != 0;
    const size_t gen = mll_fits_lds(N) ? 0 : (size_t)(pc ? (B < 128 ? B : 128) * C : B) * mll_mat_floats(N) * sizeof(float);
    if (flags & DKT_MLL_FORCE_GENERIC) return gen;
    if (N + 1 <= 128) return (flags & DKT_MLL_WANT_CHOL) && (N + 1 + 15) / 16 > 2 ? gen : 0;      // (Cholesky output beyond N = 31: the generic kernel, LDS-resident up to N ~ 190)
    if (dkt_mll_band_applies(B, C, N, flags)) return dkt_mll_band_workspace_bytes(B, C, N);
    if (!(flags & DKT_MLL_FORCE_BLOCKED) && dkt_mll_tiled_supports(N, flags, C)) return dkt_mll_tiled_workspace_bytes_form(B, C, N, pc);
    if (pc) return gen;                                                                               // N > 447 with per-class matrices: the generic kernel
    const size_t big = dkt_mll_big_workspace_bytes(B, C, N);
    return big > gen ? big : gen;
}

extern "C" int dkt_mll_f32(const float* E, const float* Y, long y_bstride, const float* sv, const float* mean,
                           const float* noise, int B, int C, int N, float jitter0, int max_tries,
                           unsigned flags, const float* cls_weight, float* logp, float* alpha, float* L,
                           float* W, float* dsv, float* dmean, float* dnoise, float* jitter_used,
                           int32_t* info, void* workspace, size_t workspace_bytes, void* stream) {
    if (!E || !Y || !sv || !mean || !noise || !logp || !alpha || !jitter_used || !info) return DKT_ERR_BAD_ARG;
    if (B <= 0 || C <= 0 || N <= 0 || max_tries < 0 || max_tries > 8 || y_bstride < 0) return DKT_ERR_BAD_ARG;
    if ((flags & DKT_MLL_WANT_GRAD) && (!W || !dsv || !dmean || !dnoise)) return DKT_ERR_BAD_ARG;
    if ((flags & DKT_MLL_WANT_CHOL) && !L) return DKT_ERR_BAD_ARG;
    MllArgs a;
    a.E = E; a.Y = Y; a.y_bstride = y_bstride; a.sv = sv; a.mean = mean; a.noise = noise;
    a.cls_weight = cls_weight; a.logp = logp; a.alpha = alpha; a.L = L; a.W = W; a.dsv = dsv;
    a.dmean = dmean; a.dnoise = dnoise; a.jitter_used = jitter_used; a.info = info;
    a.ws = (float*)workspace; a.only_failed = nullptr; a.b0 = 0; a.B = B; a.C = C; a.N = N; a.LD = mll_ld(N);
    a.jitter0 = jitter0; a.max_tries = max_tries; a.flags = flags;
    if (g_mll_env_read == 0) {
        const char* pg = dkt_variant_env("DKT_MLL_P2_GUARD");      // validation aid: a negative guard forces the grow-on-demand path of dkt_mll_h2.hip
        g_p2_guard = pg ? atoi(pg) : 1;
        const char* fm = getenv("DKT_MLL_F32MFMA");       // process-wide DKT_MLL_FORCE_F32MFMA: exact-fp32 tile products for every N <= 127 call
        g_force_f32mfma = (fm && fm[0] == '1') ? 1 : 0;
        g_mll_env_read = 1;
    }
    a.p2_guard = g_p2_guard;
    if (g_force_f32mfma && !(flags & (DKT_MLL_E_PER_CLASS | DKT_MLL_FORCE_GENERIC | DKT_MLL_FORCE_REG | DKT_MLL_FORCE_BLOCKED))) {
        flags |= DKT_MLL_FORCE_F32MFMA;
        a.flags = flags;
    }
    hipStream_t st = (hipStream_t)stream;
    if (flags & DKT_MLL_FORCE_REG) return DKT_ERR_BAD_ARG;     // (the register-sweep twin lives in libdkt_diag.so since round 4: dkt_diag_mll_reg_f32)
    if (flags & DKT_MLL_E_PER_CLASS) {
        // one base matrix per class model: the f16-split kernels with one wave per matrix (N <= 127; jitter ladder inside the kernel) or the tile-array
        // pipeline with one W per matrix (128 <= N <= 447; failed matrices redone with the jitter ladder by the generic kernel's fix-up launch);
        // DKT_MLL_FORCE_GENERIC (the validation twin) and N > 447: the generic kernel, one workgroup per matrix
        if (flags & (DKT_MLL_WANT_CHOL | DKT_MLL_FORCE_REG | DKT_MLL_FORCE_F32MFMA | DKT_MLL_FORCE_BLOCKED)) return DKT_ERR_BAD_ARG;
        if (!(flags & DKT_MLL_FORCE_GENERIC)) {
            if (N + 1 > 128 && dkt_mll_tiled_supports(N, flags, C)) {
                if (!workspace || workspace_bytes < dkt_mll_tiled_workspace_bytes_form(B, C, N, true)) return DKT_ERR_WORKSPACE;
                return dkt_mll_tiled_launch(a, workspace, workspace_bytes, st);
            }
            if (N + 1 <= 128) {
                a.kappa_max = mll_kappa_guarded(a) ? MLL_H2_KAPPA_MAX : 0.f;
                if (!dkt_mll_h2_launch(a, st)) return DKT_ERR_BAD_ARG;
                mll_kappa_fixup(a, st);
                return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
            }
        }
    }
    a.kappa_max = (N + 1 <= 128 && mll_kappa_guarded(a)) ? MLL_H2_KAPPA_MAX : 0.f;
    if (!(flags & (DKT_MLL_FORCE_GENERIC | DKT_MLL_FORCE_REG | DKT_MLL_FORCE_F32MFMA)) && dkt_mll_h2_launch(a, st)) {
        mll_kappa_fixup(a, st);
        return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    }
    if (!(flags & (DKT_MLL_FORCE_GENERIC | DKT_MLL_FORCE_REG)) && dkt_mll_mfma_launch(a, st)) return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
    // shared base matrix, many classes, 128 <= N <= 447: one orthogonal reduction per episode (dkt_mll_band.hip); DKT_MLL_FORCE_TILED = the tile-array twin
    if (dkt_mll_band_applies(B, C, N, flags) && workspace && workspace_bytes >= dkt_mll_band_workspace_bytes(B, C, N))
        return dkt_mll_band_launch(a, workspace, workspace_bytes, st);
    if (!(flags & (DKT_MLL_FORCE_GENERIC | DKT_MLL_FORCE_REG | DKT_MLL_FORCE_BLOCKED)) && dkt_mll_tiled_supports(N, flags, C) && workspace &&
        workspace_bytes >= dkt_mll_tiled_workspace_bytes_form(B, C, N, false))
        return dkt_mll_tiled_launch(a, workspace, workspace_bytes, st);
    if (!(flags & (DKT_MLL_FORCE_GENERIC | DKT_MLL_E_PER_CLASS)) && N + 1 > 128 && workspace && workspace_bytes >= dkt_mll_big_workspace_bytes(B, C, N))
        return dkt_mll_big_launch(a, workspace, workspace_bytes, st);
    const int upe = (flags & DKT_MLL_E_PER_CLASS) ? C : 1;                 // workgroups (= working matrices) per episode
    if (mll_fits_lds(N)) {
        const size_t lds = (mll_vec_floats(N) + mll_mat_floats(N)) * sizeof(float);
        if (lds > 48 * 1024) {
            if (hipFuncSetAttribute((const void*)mll_generic_kernel<false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return DKT_ERR_LAUNCH;
        }
        hipLaunchKernelGGL(mll_generic_kernel<false>, dim3((unsigned)B * upe), dim3(256), lds, st, a);
    } else {
        // global working matrices: as many episodes per launch as the caller's workspace holds
        const size_t per_ep = (size_t)upe * mll_mat_floats(N) * sizeof(float);
        const size_t fit = workspace ? workspace_bytes / per_ep : 0;
        if (fit < 1) return DKT_ERR_WORKSPACE;
        const int cnt = fit < (size_t)B ? (int)fit : B;
        for (int b0 = 0; b0 < B; b0 += cnt) dkt_mll_generic_global_launch(a, b0, (B - b0 < cnt) ? B - b0 : cnt, (float*)workspace, st);
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
