/*
 * dkt_abi.h -- C ABI of the MI355X-native DKT hot path (libdkt_hip.so, gfx950).
 *
 * The reference (BayesWatch/deep-kernel-transfer) is pure Python and has NO FFI of its own: the
 * arithmetic below is what methods/DKT.py and methods/DKT_regression.py dispatch to GPyTorch /
 * ATen (matmul, cholesky, cholesky_solve).  Each entry point cites the reference call site it
 * replaces (file:line under /root/reference); INTEGRATION.md shows the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - all matrices fp32, row-major, contiguous, batch-major [B,...], class-major [B,C,...];
 *   - every pointer is a DEVICE pointer unless named *_host; the caller (PyTorch) owns every
 *     buffer including workspace -- the library never allocates, frees or retains pointers;
 *   - hyper-parameters are device pointers (they live in torch Parameters; no host sync);
 *   - calls are asynchronous, ordered on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream), re-entrant, no global state;
 *   - return value: 0 = launched; <0 = argument error (DKT_ERR_*); failures of the numerical
 *     kind (non-positive pivot after all jitter retries) are reported per matrix in `info`
 *     on the device (LAPACK style: k+1 = index of the failing pivot), never as an exception.
 */
#ifndef DKT_ABI_H
#define DKT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DKT_ABI_VERSION 7 /* 2 (round 3): + dkt_gram_bn_train_f32, dkt_class_kernel_*, DKT_MLL_E_PER_CLASS (W layout [B,C,N,N]), DKT_MLL_FORCE_F32MFMA; \
                             3 (round 4): DKT_MLL_E_PER_CLASS up to N = 447, DKT_MLL_FORCE_REG retired (DKT_ERR_BAD_ARG), + dkt_affine_normalize_f32, dkt_normalize_bn_bwd_f32; \
                             4 (round 5): DKT_MLL_E_PER_CLASS for every N with the jitter ladder on every path (+ DKT_MLL_FORCE_GENERIC as its twin), \
                                          + dkt_predict_per_class_f32, dkt_reload_env declared, + dkt_lowrank_* (linear kernels in feature space, D <= 64 < N); \
                             5 (round 6): shared-E calls with 12 <= C <= 32 classes, 128 <= N <= 432 and >= 192 episodes take ONE band reduction per episode (dkt_mll_band.hip); \
                                          + DKT_MLL_FORCE_TILED (the tile-array kernels as its twin), DKT_MLL_FORCE_BAND; the f16-split kernels (N <= 127) are followed by a kappa-aware fix-up launch (DKT_MLL_NO_KAPPA_GUARD); \
                             6 (round 6): + dkt_objective_f32, dkt_hyper_grads_f32; \
                             7 (round 6): + dkt_bn_param_grads_f32 (+ its workspace query); the kappa test of the f16-split kernels moved into those kernels (one launch less per call) */

/* status codes */
#define DKT_OK 0
#define DKT_ERR_BAD_ARG (-1)
#define DKT_ERR_TOO_LARGE (-2)
#define DKT_ERR_WORKSPACE (-3)
#define DKT_ERR_LAUNCH (-4)

/* base-kernel kinds (configs.py:7 kernel_type; DKT.py:352-370) */
#define DKT_KERNEL_LINEAR 0 /* linear / cossim / bncossim : E = A B^T                       */
#define DKT_KERNEL_RBF 1    /* rbf : E = exp(-0.5 |a-b|^2 / l^2), centred norm expansion    */
#define DKT_KERNEL_SQDIST 2 /* U = |a-b|^2 / l^2 (clamped >= 0): building block of matern   */
#define DKT_KERNEL_LINEAR_UNIT 3 /* LINEAR + the caller's promise |a_ik| <= 1 for every element (cossim / bncossim:
                                    the features went through F.normalize, methods/DKT.py:141-142).  Same result to fp32
                                    accuracy; lets the episode-resident kernel use the scaled 2-way f16 split.  An element
                                    beyond 1.999 overflows to inf / NaN in that episode's output (loud, not silent). */

/* gram_bwd flags */
#define DKT_GRAM_UNIT_ROWS 1u /* the same promise for the rows of Z in dkt_gram_bwd_f32 (W is unrestricted) */
#define DKT_GRAM_W_SYMMETRIC 2u /* dkt_gram_bwd_f32: the caller states that every W[b] is symmetric (dkt_mll_f32 writes it so); */
                                /* s (W + W^T) is then formed as 2 s W from row reads only (used by the 128 < N <= 448 kernel)    */

/* mll flags */
#define DKT_MLL_WANT_GRAD 1u /* also produce W (d obj / d E) and the per-class hyper grads */
#define DKT_MLL_WANT_CHOL 2u /* also write the Cholesky factors L[B,C,N,N] (what dkt_predict_var_f32 / confidence_region need: the regression head conditions on
                                5 .. 19 rows, DKT_regression.py:84-93).  Served by the exact-fp32 MFMA kernel for N <= 31, the generic kernel for 32 <= N <= 127, the
                                blocked path above */
#define DKT_MLL_FORCE_GENERIC 4u /* validation aid: take the generic LDS/global path for any N  */
#define DKT_MLL_FORCE_REG 8u     /* RETIRED in ABI 3 (dkt_mll_f32 answers DKT_ERR_BAD_ARG): the round-1 register-sweep kernel is a validation twin in the
                                    measurement library now (libdkt_diag.so: dkt_diag_mll_reg_f32), not part of the product */
#define DKT_MLL_FORCE_BLOCKED 16u /* validation aid: the blocked batched-GEMM path instead of the tile-array kernels (N > 127) */
#define DKT_MLL_FORCE_F32MFMA 32u /* exact-fp32 arithmetic instead of the f16-split kernels (N <= 127; no range contract, see below).  The product library serves it
                                     with its generic exact-fp32 kernel; the exact-fp32 MFMA twin of the split kernels lives in the twins library */
#define DKT_MLL_FORCE_TILED 128u /* validation aid (ABI 5): the tile-array kernels (one factorisation per class matrix) instead of the shared band reduction of
                                    dkt_mll_band.hip, which is the default for a shared base matrix with 12 <= C <= 32 classes, 128 <= N <= 432 and >= 192 episodes per call */
#define DKT_MLL_FORCE_BAND 256u  /* validation / measurement aid (ABI 5): the band reduction wherever it is defined (shared base matrix, 128 <= N <= 432, 2 <= C <= 32), also
                                    outside the window in which it is the default */
#define DKT_MLL_NO_KAPPA_GUARD 512u /* validation / measurement aid (ABI 5): N <= 127 only -- the raw f16-split kernels without the fix-up launch that hands a unit whose
                                       a-priori condition bound 1 + sv trace(E) / noise exceeds 5e3 to the exact-fp32 generic kernel */
#define DKT_MLL_E_PER_CLASS 64u  /* every class model has its OWN base matrix: E is [B,C,N,N] and W is [B,C,N,N] (no sum over the classes) */

int dkt_abi_version(void);

/* Two builds of this ABI exist.  libdkt_hip.so, the PRODUCT: one kernel per (call, shape class), no measurement switch -- the only environment variables it reads
 * are the dispatch thresholds DKT_GRAM_EP_MINB, DKT_MLL_H2E_MINB, DKT_MLL_TILED_CHUNK and the process-wide exact-fp32 request DKT_MLL_F32MFMA=1.
 * libdkt_twins.so (the same sources compiled with -DDKT_TWINS): in addition every pipeline variant, legacy pipeline and validation twin the defaults were
 * measured against, selected by the environment switches of DESIGN.md's appendix; loaded by the test-suite and the A/B tools only. */

/* The library reads its measurement / validation switches (environment variables) once, at the first call that needs
 * them; a host that changed one inside a running process (tests, A/B tools) calls this to have them re-read.  No effect on results of the
 * default configuration.  (Exported since round 2; declared here since ABI 4.)
 * Threads: every entry point is re-entrant on its arguments (no buffers, handles or streams are kept between calls); the only process-wide state is this cache of the
 * environment's dispatch thresholds -- plain ints, filled lazily and idempotently (concurrent first calls store the same value), which the environment being per-process
 * makes process-wide by nature.  dkt_reload_env() itself is for single-threaded harnesses: do not call it while another thread is inside a dkt_* call. */
void dkt_reload_env(void);

/* Device query used by the host side to fail loudly on a non-gfx950 box: returns the number of
 * compute units of the current HIP device (>0) or <0 on error. */
int dkt_device_cu_count(void);

/*
 * dkt_gram_f32 -- base kernel matrix of one or many episodes.
 *   E[b] = k(A[b], Bm[b])  with A:[B,M,D], Bm:[B,N,D] -> E:[B,M,N].
 *   Bm == NULL: symmetric Gram of A with itself (M == N, only lower tiles computed, mirrored).
 *   kind = DKT_KERNEL_LINEAR | DKT_KERNEL_LINEAR_UNIT | DKT_KERNEL_RBF | DKT_KERNEL_SQDIST; `lengthscale` (device, 1 float)
 *   unused by the linear kinds.
 * Replaces: ExactGPLayer.forward -> covar_module(x) (methods/DKT.py:375-378,
 *   methods/DKT_regression.py:126-129), i.e. GPyTorch LinearKernel / RBFKernel evaluation,
 *   evaluated once per episode instead of once per class model (DKT.py:148-149,161).
 */
int dkt_gram_f32(const float* A, const float* Bm, float* E, int B, int M, int N, int D, int kind,
                 const float* lengthscale, void* stream);

/*
 * dkt_mll_workspace_bytes -- bytes of scratch dkt_mll_f32 needs for (B,C,N) (0 when the
 * factorisation is LDS/register resident).
 */
size_t dkt_mll_workspace_bytes(int B, int C, int N);
/* The same for ONE call: the bytes dkt_mll_f32 needs with exactly these flags (never more than dkt_mll_workspace_bytes(B, C, N), which covers every
 * flag combination; a default shared-matrix call of a few 420-row episodes needs a tenth of it).  (ABI 4) */
size_t dkt_mll_workspace_bytes_for(int B, int C, int N, unsigned flags);

/*
 * dkt_mll_f32 -- exact-GP marginal log likelihood of C one-vs-rest models per episode that
 * share the episode's base matrix E[b]:  K_c = sv[c] * E[b] + noise[c] * I.
 *   jittered Cholesky (try 0, then total diagonal jitter jitter0 * 10^i, i < max_tries),
 *   log-det, r = y_c - mean[c], alpha = K^-1 r,
 *   logp[b,c] = -0.5 (r^T alpha + logdet K + N log 2 pi).
 *   Y: [*,C,N] targets; episode b reads Y + b * y_bstride (y_bstride = 0: shared targets).
 *   sv, mean, noise: [C] device (outputscale*variance, ConstantMean, likelihood noise).
 * Outputs (device): logp[B,C], alpha[B,C,N], jitter_used[B,C], info[B,C] (0 = ok).
 *   flags & DKT_MLL_WANT_CHOL : L[B,C,N,N] lower factors (upper part zero).
 *   flags & DKT_MLL_WANT_GRAD : with M_c = 0.5 (alpha alpha^T - K_c^-1):
 *        W[b]      = sum_c cls_weight[c] * sv[c] * M_c      (= d obj_b / d E[b], symmetric)
 *        dsv[b,c]  = sum_ij M_c,ij E_ij   dmean[b,c] = sum_i alpha_i   dnoise[b,c] = tr M_c
 *        (raw d logp[b,c] / d theta_c; the caller applies cls_weight / upstream grads)
 *        cls_weight: [C] device or NULL (= 1).
 *   flags & DKT_MLL_E_PER_CLASS : the class models do not share a base matrix (rbf / matern / polynomial kernels with per-class
 *        lengthscale / offset: one ExactGPLayer per class, methods/DKT.py:63-66, 352-365):  K_c = sv[c] * E[b,c] + noise[c] * I with
 *        E:[B,C,N,N], and W:[B,C,N,N] holds W[b,c] = cls_weight[c] * sv[c] * M_c = d obj_b / d E[b,c] per class.  One launch for all
 *        classes, every N: N <= 127 the f16-split kernels with one wave per matrix (jitter ladder inside the kernel), 128 <= N <= 447 the
 *        tile-array pipeline with one W per matrix (needs the workspace of dkt_mll_workspace_bytes; a matrix that fails attempt 0 is redone
 *        with psd_safe_cholesky's jitter ladder by a fix-up launch of the generic kernel, as for a shared E), N > 447 the generic kernel
 *        (one workgroup per matrix; working matrices in the workspace).  DKT_MLL_FORCE_GENERIC selects the generic kernel for any N (the
 *        validation twin); DKT_MLL_WANT_CHOL and the other DKT_MLL_FORCE_* do not combine with it (DKT_ERR_BAD_ARG).
 * Range contract of the default kernels for N <= 127 (scaled 2-way f16 splits on the f16 matrix pipe, 22 significand bits, fp32
 *   accumulate): every K_c = sv[c] E + noise[c] I must satisfy |K_ij| <= max_i K_ii, which every positive semi-definite E (any Gram /
 *   RBF / Matern / polynomial base matrix) does.  An E that violates it (not PSD, or user-supplied with off-diagonals beyond the
 *   diagonal) can overflow the f16 planes: the matrix then reports info != 0 / NaN outputs -- loud, never a silently wrong value.
 *   DKT_MLL_FORCE_F32MFMA (per call) or the environment variable DKT_MLL_F32MFMA=1 (process-wide, read at the first call and at
 *   dkt_reload_env()) selects the exact-fp32 twin, which has no such contract.
 * Replaces: `loss = -self.mll(output, self.model.train_targets)` and its autograd backward
 *   (methods/DKT.py:161-163, 252-254; methods/DKT_regression.py:53-56): GPyTorch
 *   GaussianLikelihood.marginal + MultivariateNormal.log_prob + psd_safe_cholesky +
 *   cholesky_solve, and the mean cache of DefaultPredictionStrategy (DKT.py:177,187,265,330).
 */
int dkt_mll_f32(const float* E, const float* Y, long y_bstride, const float* sv, const float* mean,
                const float* noise, int B, int C, int N, float jitter0, int max_tries,
                unsigned flags, const float* cls_weight, float* logp, float* alpha, float* L,
                float* W, float* dsv, float* dmean, float* dnoise, float* jitter_used,
                int32_t* info, void* workspace, size_t workspace_bytes, void* stream);

/*
 * dkt_gram_bwd_f32 -- backward of the linear Gram: dZ[b] = scale_b * (W[b] + W[b]^T) Z[b].
 *   W:[B,N,N]  Z:[B,N,D]  dZ:[B,N,D];  ep_scale: [B] device per-episode factor (upstream
 *   gradient of the episode objective) or NULL (= 1).  flags: 0 or DKT_GRAM_UNIT_ROWS | DKT_GRAM_W_SYMMETRIC.
 * Replaces: autograd through matmul(Z, Z^T) in loss.backward() (methods/DKT.py:163).
 */
int dkt_gram_bwd_f32(const float* W, const float* Z, float* dZ, int B, int N, int D,
                     const float* ep_scale, unsigned flags, void* stream);

/*
 * dkt_rbf_bwd_f32 -- chain rule of the RBF base kernel.  With Ws = 0.5 (W + W^T) and
 *   A = -(Ws o E) / l^2 :  Wp[b] = diag(A 1) - A  (so that dZ = 2 Wp Z = dkt_gram_bwd_f32(Wp, Z)),
 *   dlengthscale[b] = sum_ij Ws_ij E_ij d2_ij / l^3,  d2_ij = -2 l^2 ln E_ij.
 * Replaces: autograd through RBFKernel.forward in loss.backward() (methods/DKT_regression.py:56).
 */
int dkt_rbf_bwd_f32(const float* W, const float* E, const float* lengthscale, float* Wp,
                    float* dlengthscale, int B, int N, void* stream);

/*
 * dkt_objective_f32 -- the episode's training objective from its class models' log marginal likelihoods:
 *   obj[b] = sum_c cls_weight[c] * logp[b,c]           (cls_weight NULL = 1)
 * dkt_hyper_grads_f32 -- chain rule from dkt_mll_f32's raw per-episode, per-class gradients to the [C] parameters:
 *   g_x[c] = cls_weight[c] * sum_b gobj[b] * d_x[b,c]      x in {sv, mean, noise}; a NULL (d_x, g_x) pair is skipped.
 * Both in a fixed summation order (bitwise reproducible).  As tensor expressions these are seven launches per training step; the
 * reference's loop is one episode per step (methods/DKT.py:117-164) and bound by launches.
 * Replaces: the sum over the class models in `loss = -self.mll(output, self.model.train_targets)` (SumMarginalLogLikelihood over the
 *   IndependentModelList, methods/DKT.py:70-71, 161) and autograd's accumulation into the C models' outputscale / mean / noise
 *   parameters (DKT.py:163).
 */
int dkt_objective_f32(const float* logp, const float* cls_weight, float* obj, int B, int C, void* stream);
int dkt_hyper_grads_f32(const float* gobj, const float* cls_weight, const float* dsv, const float* dmean, const float* dnoise,
                        float* gsv, float* gmean, float* gnoise, int B, int C, void* stream);

/*
 * dkt_sqdist_bwd_f32 -- chain rule of U = |z_i - z_j|^2 / l^2 (DKT_KERNEL_SQDIST; MaternKernel, DKT.py:358-359):
 *   with Ws = 0.5 (W + W^T) (W = d obj / d U):  Wp = diag(A 1) - A, A = 2 Ws / l^2  (dZ = dkt_gram_bwd_f32(Wp, Z)),
 *   dlengthscale[b] = -(2 / l) sum_ij Ws_ij U_ij.
 */
int dkt_sqdist_bwd_f32(const float* W, const float* U, const float* lengthscale, float* Wp,
                       float* dlengthscale, int B, int N, void* stream);

/*
 * dkt_predict_f32 -- posterior mean of the C models at M test points and the arg-max label.
 *   Ex:[B,M,N] base cross kernel k(x*, X_cond); alpha:[B,C,N]; sv, mean:[C] device.
 *   mu[b,c,q] = mean[c] + sv[c] * sum_n Ex[b,q,n] alpha[b,c,n];   labels[b,q] = argmax_c mu
 *   (first maximum wins, as np.argmax over sigmoid(mu) does -- sigmoid is monotone).
 *   labels may be NULL.
 * Replaces: `self.likelihood(*self.model(*z_query_list))` + sigmoid + vstack + argmax
 *   (methods/DKT.py:176-181, 186-191, 264-270, 329-334; DKT_regression.py:92).
 */
int dkt_predict_f32(const float* Ex, const float* alpha, const float* sv, const float* mean,
                    float* mu, int32_t* labels, int B, int C, int M, int N, void* stream);

/*
 * dkt_predict_per_class_f32 -- the same with one base cross kernel PER CLASS MODEL (rbf / matern / polynomial kernels whose class models own
 *   their lengthscale / offset, methods/DKT.py:63-66, 352-365):  Ex:[B,C,M,N],  mu[b,c,q] = mean[c] + sv[c] * sum_n Ex[b,c,q,n] alpha[b,c,n].
 * Replaces: the same reference lines as dkt_predict_f32 for kernel_type in {rbf, matern, poli1, poli2}.
 */
int dkt_predict_per_class_f32(const float* Ex, const float* alpha, const float* sv, const float* mean,
                              float* mu, int32_t* labels, int B, int C, int M, int N, void* stream);

/*
 * dkt_predict_var_f32 -- diagonal of the predictive covariance (what confidence_region() reads,
 *   methods/DKT_regression.py:92-93):  var[b,c,q] = sv_c exx[b,q] - sv_c^2 |L_c^-1 Ex[b,q,:]|^2 + noise_c.
 *   L:[B,C,N,N] from dkt_mll_f32(DKT_MLL_WANT_CHOL); exx:[B,M] = k(x*,x*) diagonal of the base kernel.
 */
int dkt_predict_var_f32(const float* Ex, const float* exx, const float* L, const float* sv,
                        const float* noise, float* var, int B, int C, int M, int N, void* stream);

/*
 * ---- Linear kernels in FEATURE space: D <= 64 < N (round 5; Conv4S / Omniglot, backbone.py:287-310, train.py:85-93, 132) ----------
 * K_c = sv_c Z Z^T + noise_c I has rank D + the noise floor: with A = Z^T Z, p_c = Z^T (y_c - m_c) and K'_c = sv_c A + noise_c I (D x D)
 *     log det K_c = (N - D) log noise_c + log det K'_c,   alpha_c = K_c^-1 r_c = (r_c - sv_c Z t_c) / noise_c,   t_c = K'_c^-1 p_c = Z^T alpha_c,
 *     d obj / d Z = sum_c cw_c sv_c (alpha_c - Z t_c) t_c^T + 2 Z W',   W' = 0.5 sum_c cw_c sv_c (t_c t_c^T - K'_c^-1),
 * and K'_c is the matrix family dkt_mll_f32 factorises (base matrix A, targets p_c, zero mean, the same sv / noise / cls_weight / jitter ladder --
 * the jitter lands on noise_c exactly as psd_safe_cholesky's does on K_c's diagonal).  The training episode is then
 *     dkt_lowrank_noise_floor_f32 -> dkt_lowrank_gram_f32 -> dkt_mll_f32 (B, C, N' = DKT_LOWRANK_DP, y_bstride = C * DKT_LOWRANK_DP, mean = 0, DKT_MLL_WANT_GRAD) -> dkt_lowrank_finish_f32
 *     backward: dkt_lowrank_bwd_f32
 * and neither E[B,N,N] nor W[B,N,N] exists.  Z:[B,N,D] row-major, D % 4 == 0, D <= DKT_LOWRANK_DP (zero-padded to it), C <= 32, any N
 * (DKT_ERR_TOO_LARGE otherwise: the caller takes dkt_gram_f32 / dkt_mll_f32 / dkt_gram_bwd_f32).  A:[B,DP,DP], P and T:[B,C,DP], Wd:[B,DP,DP] hold the
 * D x D problem in a FIXED PERMUTED feature order (index 16 q + m stands for column 4 m + q), private to these four calls.
 * Replaces, for these shapes: methods/DKT.py:375-378 (LinearKernel), 161-163 (marginal likelihood + backward) -- the same lines as the three
 * N x N kernels; results identical to them to fp32 accuracy (tests: both paths against the float64 oracle, and against each other).
 */
#define DKT_LOWRANK_DP 64
int dkt_lowrank_supported(int C, int N, int D); /* 1 when the three calls below accept (C, N, D) */
/* A[b] = Zp^T Zp (exactly symmetric), P[b,c,:] = Zp^T (Y[b,c,:] - mean[c]);  Y: [*,C,N], episode b reads Y + b * y_bstride */
int dkt_lowrank_gram_f32(const float* Z, const float* Y, long y_bstride, const float* mean, float* A, float* P,
                         int B, int C, int N, int D, void* stream);
/* Which rung of psd_safe_cholesky's ladder (TOTAL jitter 0, jitter0, 10 jitter0, ...) the N x N matrix K_c needs before it is numerically positive definite at
 * all: with D < N it has N - D eigenvalues equal to noise_c, so its fp32 factorisation fails -- and the reference retries -- when noise_c + jitter < 2^-22 max_i K_ii
 * = 2^-22 (sv_c zmax2 + noise_c).  The D x D models never see that rank deficiency, so the rung is chosen here ([C] device arithmetic, no read-back):
 * pre_jitter[c] (NaN when no rung clears the floor: the class then fails), noise_eff[c] = noise[c] + pre_jitter[c] = the noise the other three calls and the
 * D x D dkt_mll_f32 (whose own ladder runs on top) are given.  zmax2: device scalar max_i |z_i|^2 over the call's rows, or NULL (= 1: rows that went through F.normalize). */
int dkt_lowrank_noise_floor_f32(const float* sv, const float* noise, const float* zmax2, float jitter0, int max_tries, float* noise_eff,
                                float* pre_jitter, int C, void* stream);
/* from the D x D call's outputs T = alpha', logp_d = logp', dnoise_d = dnoise', jitter_used (noise = noise_eff):  logp[B,C], alpha[B,C,N], dsv / dmean / dnoise [B,C]
 * (the conventions of dkt_mll_f32), V[B,C,N] = cls_weight_c sv_c (alpha_c - Z t_c) for the backward, and optionally obj[B] = sum_c cls_weight_c logp[b,c] and
 * jitter_total[B,C] = jitter_used + pre_jitter (pre_jitter / jitter_total / obj may be NULL).  A failed class (NaN in T / logp_d) comes out NaN. */
int dkt_lowrank_finish_f32(const float* Z, const float* Y, long y_bstride, const float* sv, const float* mean, const float* noise,
                           const float* cls_weight, const float* T, const float* logp_d, const float* dnoise_d, const float* jitter_used,
                           const float* pre_jitter, float* jitter_total, float* obj,
                           float* logp, float* alpha, float* V, float* dsv, float* dmean, float* dnoise,
                           int B, int C, int N, int D, void* stream);
/* dZ[b] = ep_scale[b] * (V[b]^T T[b] + 2 Z[b] Wd[b])  (= ep_scale[b] (W + W^T) Z of the N x N formulation);  ep_scale: [B] device or NULL (= 1) */
int dkt_lowrank_bwd_f32(const float* Z, const float* V, const float* T, const float* Wd, const float* ep_scale, float* dZ,
                        int B, int C, int N, int D, void* stream);

/*
 * ---- BNCosSim front half fused into the Gram build ------------------------------------
 * Reference: bn_out = nn.BatchNorm1d(D) appended to the trunk (methods/DKT.py:48), z = F.normalize(trunk(x), p=2, dim=1)
 * (DKT.py:141-142, 174-175, 236-237), LinearKernel on z (DKT.py:375-378).  X is the trunk output BEFORE bn_out.
 *
 * dkt_bn_stats_f32 -- train-mode batch statistics per episode and feature (the "batch" of bn_out is the N images of
 *   one episode): mean[b,d], rstd[b,d] = 1/sqrt(biased var + eps), the folded affine map  y = a x + s  with
 *   a = gamma rstd, s = beta - mean a, and the unbiased variance torch feeds the running estimate (may be NULL).
 *   gamma / beta may be NULL (1 / 0).  D % 4 == 0, X 16-byte aligned.
 */
int dkt_bn_stats_f32(const float* X, const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                     float* a, float* s, float* var_unbiased, int B, int N, int D, void* stream);

/*
 * dkt_gram_bn_f32 -- E[b] = Zn Zn^T,  Zn_i = y_i / max(||y_i||_2, 1e-12),  y = a x + s  applied while X is staged
 *   (Zn is never written).  a, s: [B,D] with ab_bstride = D (train-mode statistics of dkt_bn_stats_f32) or [D] with
 *   ab_bstride = 0 (eval mode: a = gamma / sqrt(running_var + eps), s = beta - running_mean a; a = 1, s = 0 for the
 *   plain cossim kernel).  rnorm[b,i] = 1 / max(||y_i||, 1e-12) is returned for the backward.  N <= 128, D % 4 == 0.
 */
int dkt_gram_bn_f32(const float* X, const float* a, const float* s, long ab_bstride, float* E, float* rnorm,
                    int B, int N, int D, void* stream);

/*
 * dkt_gram_bn_train_f32 -- dkt_bn_stats_f32 + dkt_gram_bn_f32 in ONE pass over X (round 3): the train-mode batch statistics of a
 *   32-feature slice are taken while all N rows of the slice sit in the workgroup's registers on their way into LDS, folded into
 *   a / s and applied to the same registers.  Outputs as dkt_bn_stats_f32 (mean, rstd, a, s, var_unbiased: each [B,D];
 *   var_unbiased may be NULL) and dkt_gram_bn_f32 (E, rnorm).  gamma / beta: [D] or NULL (1 / 0).  N <= 128, D % 4 == 0.
 *   Round 5: at N > 32 the products run as a scaled 2-way f16 split whose scale comes from train-mode BatchNorm's a-priori bound
 *   |y| <= |beta| + |gamma| sqrt(N - 1); an episode with a row whose norm is too far below that bound for the split (checked on the device) is
 *   recomputed by the 3-way bf16 kernel in a second launch of the same call -- two launches on `stream`, results to fp32 accuracy either way.
 *   Replaces bn_out in train mode + F.normalize + LinearKernel (methods/DKT.py:48, 141-142, 375-378) of a training episode.
 */
int dkt_gram_bn_train_f32(const float* X, const float* gamma, const float* beta, float eps, float* mean, float* rstd,
                          float* a, float* s, float* var_unbiased, float* E, float* rnorm, int B, int N, int D, void* stream);

/*
 * dkt_gram_bn_bwd_f32 -- backward of dkt_gram_bn_f32 (+ the batch-statistics dependence of train-mode BatchNorm1d):
 *   given W[b] = d obj / d E[b] and the per-episode upstream scale ep_scale[b] (NULL: 1), returns
 *   dX[B,N,D] = d obj / d X and, when mean/rstd are given (train mode), the per-episode parts
 *   dgamma_part[b,d] = sum_i dY_id xhat_id, dbeta_part[b,d] = sum_i dY_id (the caller sums over b).
 *   mean == NULL: the affine map is treated as constant (eval mode / no bn_out), dgamma_part / dbeta_part untouched.
 * Replaces autograd through matmul, F.normalize and BatchNorm1d (loss.backward(), methods/DKT.py:163).
 */
int dkt_gram_bn_bwd_f32(const float* W, const float* E, const float* X, const float* a, const float* s, long ab_bstride,
                        const float* mean, const float* rstd, const float* rnorm, const float* ep_scale, float* dX,
                        float* dgamma_part, float* dbeta_part, int B, int N, int D, void* stream);

/*
 * dkt_bn_param_grads_f32 -- the sum over the episodes of a call that dkt_gram_bn_bwd_f32 / dkt_normalize_bn_bwd_f32 leave to the caller:
 *   dgamma[d] = sum_b dgamma_part[b,d],  dbeta[d] = sum_b dbeta_part[b,d]      (parts [B,D], D % 4 == 0, 16-byte aligned)
 *   in a fixed summation order (row lanes of 256-row chunks in a tree, then the chunks in order: bitwise reproducible), one launch up to 256 episodes, two
 *   beyond (workspace: dkt_bn_param_grads_workspace_bytes(B, D) bytes, 0 up to 256 episodes).  As two tensor reductions this is two launches plus, on ROCm,
 *   four buffer fills for their cross-block semaphores -- 40 us of a 1.3-ms step of 2048 episodes.
 * Replaces autograd's accumulation into bn_out.weight / bn_out.bias (loss.backward(), methods/DKT.py:48, 163) for a batch of episodes.
 */
size_t dkt_bn_param_grads_workspace_bytes(int B, int D);
int dkt_bn_param_grads_f32(const float* dgamma_part, const float* dbeta_part, float* dgamma, float* dbeta, int B, int D,
                           void* workspace, size_t workspace_bytes, void* stream);

/*
 * ---- the same front end for episodes of MORE than 128 rows (round 4; the 20-way shapes of train.py:132-133) -----------------------
 * The large-N Gram kernels take unit rows as their input, so Zn is written once:
 *   forward : dkt_bn_stats_f32 (train mode; or the eval-mode / identity a, s)  ->  dkt_affine_normalize_f32  ->  dkt_gram_f32(DKT_KERNEL_LINEAR_UNIT)
 *   backward: dkt_gram_bwd_f32 (dZn)  ->  dkt_normalize_bn_bwd_f32
 *
 * dkt_affine_normalize_f32 -- Zn[b,i,:] = y / max(||y||_2, 1e-12), y = a x + s; rnorm[b,i] = 1 / max(||y||, 1e-12).  a, s: [B,D] (ab_bstride = D) or [D]
 *   (ab_bstride = 0).  Any N; D % 4 == 0.  Replaces bn_out (affine part) + F.normalize (methods/DKT.py:48, 141-142).
 * dkt_normalize_bn_bwd_f32 -- given dZn = d obj / d Zn: dX = d obj / d X through F.normalize and the affine map, and -- when mean / rstd are given (train-mode
 *   BatchNorm1d: the statistics depend on X) -- through the batch statistics too, with the per-episode parts dgamma_part[b,d] = sum_i dY_id xhat_id,
 *   dbeta_part[b,d] = sum_i dY_id (the caller sums over b).  mean == NULL: the affine map is a constant (eval mode / no bn_out; X, rstd, dgamma_part, dbeta_part
 *   unused).  a: [B,D] (a_bstride = D) or [D] (0).  rowdot_ws: [B,N] floats of scratch.  N <= 1024, D % 4 == 0.
 *   Replaces autograd through F.normalize and BatchNorm1d (loss.backward(), methods/DKT.py:163).
 */
int dkt_affine_normalize_f32(const float* X, const float* a, const float* s, long ab_bstride, float* Zn, float* rnorm, int B, int N, int D, void* stream);
int dkt_normalize_bn_bwd_f32(const float* dZn, const float* Zn, const float* X, const float* a, long a_bstride, const float* mean, const float* rstd,
                             const float* rnorm, float* dX, float* dgamma_part, float* dbeta_part, float* rowdot_ws, int B, int N, int D, void* stream);

/*
 * ---- per-class element-wise maps of the non-linear base kernels ------------------------------
 * Reference: one ExactGPLayer per class, each with its own RBFKernel / MaternKernel(nu = 2.5) lengthscale or PolynomialKernel offset
 * (methods/DKT.py:63-66, 352-365).  All C kernels of an episode are functions of ONE contraction (dkt_gram_f32: DKT_KERNEL_SQDIST with
 * lengthscale 1, or DKT_KERNEL_LINEAR):
 *
 * dkt_class_kernel_f32 -- E[b,c,k] = f(base[b,k]; param[c]), k < NN (NN = N*N, or M*N for a cross kernel):
 *   DKT_CLASSMAP_RBF       u = base / l_c^2, f = exp(-u / 2)
 *   DKT_CLASSMAP_MATERN25  u = base / l_c^2, r = sqrt(5 max(u, 1e-30)), f = (1 + r + r^2 / 3) exp(-r)
 *   DKT_CLASSMAP_POLY      f = (base + offset_c)^power, power = 1 or 2
 *
 * dkt_class_kernel_bwd_f32 -- chain rule behind the marginal-likelihood launch with DKT_MLL_E_PER_CLASS: W[B,C,N,N] = d obj / d E (symmetric
 *   per matrix: what dkt_mll_f32 writes) -> Wp[B,N,N] such that d obj / d Z = dkt_gram_bwd_f32(Wp, Z) = (Wp + Wp^T) Z
 *   (distance kinds: Wp = diag(A 1) - A, A = 2 d obj / d d2; POLY: Wp = d obj / d (z_i . z_j)), and dparam[b,s,c] = the part of d obj / d l_c (or
 *   offset_c) of episode b that row split s contributes: dparam is [B, nsplit, C] with nsplit = dkt_class_kernel_bwd_nsplit(B, N) (1 for batches that
 *   fill the GPU; small batches split the rows of an episode over several workgroups); the caller sums over s (and b).  C <= 32.
 * Replaces autograd through the C kernel evaluations (loss.backward(), methods/DKT.py:163).
 */
#define DKT_CLASSMAP_RBF 0
#define DKT_CLASSMAP_MATERN25 1
#define DKT_CLASSMAP_POLY 2
int dkt_class_kernel_f32(const float* base, int kind, const float* param, int power, float* E, int B, int C, int NN, void* stream);
int dkt_class_kernel_bwd_nsplit(int B, int N);
int dkt_class_kernel_bwd_f32(const float* W, const float* base, int kind, const float* param, int power, float* Wp, float* dparam,
                             int B, int C, int N, void* stream);

/*
 * dkt_smk_f32 -- spectral-mixture base kernel of the regression head
 *   (gpytorch.kernels.SpectralMixtureKernel(num_mixtures=4, ard_num_dims=2916), methods/DKT_regression.py:121-122):
 *   E[b,i,j] = sum_q weights[q] prod_d exp(-2 pi^2 (scales[q,d] tau_d)^2) cos(2 pi means[q,d] tau_d),
 *   tau = x1[b,i,:] - x2[b,j,:].   x1:[B,M,D]; x2:[B,N,D] or NULL (symmetric, M == N); weights:[Q]; means, scales:[Q,D]
 *   (constrained values); Q <= 8.  Eq[B,Q,M,N] (nullable) receives the per-mixture terms the backward needs.
 */
int dkt_smk_f32(const float* x1, const float* x2, const float* weights, const float* means, const float* scales,
                float* E, float* Eq, int B, int M, int N, int D, int Q, void* stream);

/*
 * dkt_smk_bwd_f32 -- chain rule of the symmetric spectral-mixture matrix: given gE[b] = d obj / d E[b] (any, not
 *   necessarily symmetric) and Eq of the forward, returns dx[B,N,D] and the per-episode parts dmeans[B,Q,D],
 *   dscales[B,Q,D] (the caller sums over b; d obj / d weights[q] = sum_bij gE Eq is an element-wise reduction left to
 *   the caller).  N <= 232.  Replaces autograd through the kernel's forward (loss.backward(), DKT_regression.py:56).
 */
int dkt_smk_bwd_f32(const float* gE, const float* Eq, const float* x, const float* weights, const float* means,
                    const float* scales, float* dx, float* dmeans, float* dscales, int B, int N, int D, int Q,
                    void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DKT_ABI_H */
