// dkt_lowrank.hip -- the training episode of the LINEAR kernels (linear / cossim / bncossim) in FEATURE space when the feature
// dimension is smaller than the episode: D <= 64 < N (Omniglot's Conv4S trunk, backbone.py:287-310: D = 64; train.py:132: N = 105
// 5-way, 420 20-way).
//
// K_c = sv_c Z Z^T + noise_c I (N x N) has rank D + the noise floor, so nothing N x N has to be built, factorised or inverted
// (replaces, for these shapes, the same reference lines as dkt_gram_f32 + dkt_mll_f32 + dkt_gram_bwd_f32: methods/DKT.py:375-378,
// 161-163 -- GPyTorch's LinearKernel would itself keep a low-rank root here).  With A = Z^T Z (D x D), p_c = Z^T r_c, r_c = y_c - m_c and
// the D x D model  K'_c = sv_c A + noise_c I :
//     det K_c       = noise_c^(N - D) det K'_c                               (Sylvester / Weinstein-Aronszajn)
//     K_c^-1 r      = (r - sv_c Z t_c) / noise_c,   t_c = K'_c^-1 p_c        (Woodbury)
//     Z^T alpha_c   = t_c                                                    (push-through identity)
//     r^T K_c^-1 r  = (|r|^2 - sv_c p_c^T t_c) / noise_c
//     tr K_c^-1     = (N - D) / noise_c + tr K'_c^-1
//     d obj / d Z   = (W + W^T) Z = sum_c cw_c sv_c (alpha_c - Z t_c) t_c^T + 2 Z W',   W' = 0.5 sum_c cw_c sv_c (t_c t_c^T - K'_c^-1)
// and K'_c = sv_c A + noise_c I is EXACTLY the matrix family dkt_mll_f32 factorises (base matrix A, targets p_c, zero mean, the same
// outputscale / noise, the same class weights): its jittered Cholesky adds psd_safe_cholesky's jitter to noise_c as the reference does,
// its alpha output is t_c, its W output is W', its d/dnoise output gives tr K'_c^-1.  So the episode is
//     dkt_lowrank_gram_f32   : A[b] = Z^T Z, P[b] = Z^T R                        (one pass over Z)
//     dkt_mll_f32            : on (A, P) with N' = 64 instead of N               (5 x 5 tiles instead of 7 x 7 at N = 105, 27 x 27 at N = 420)
//     dkt_lowrank_finish_f32 : alpha, logp, the hyper-parameter gradients, V = cw sv (alpha - Z t)   (second pass over Z)
//     dkt_lowrank_bwd_f32    : dZ = g_b (V T^T + 2 Z W')                         (backward: third pass over Z, dZ written once)
// E[B,N,N] and W[B,N,N] never exist (44 KB + 44 KB per cfg1 episode, 706 KB + 706 KB at N = 420).
//
// All three kernels: one wave per episode, v_mfma_f32_16x16x4_f32 (exact fp32 products, no operand splitting), operands straight
// from global memory into MFMA registers.  Feature order: a lane's 16-byte load holds columns 4 m .. 4 m + 3 of a row, and register q
// of every lane forms operand tile q, i.e. feature index d' = 16 q + m stands for column 4 m + q.  A, P, t and W' live in that
// (fixed) permuted order -- a simultaneous row / column permutation of the D x D problem, invisible outside these kernels -- which makes
// every Z load and every dZ store a full 256-byte row segment.  D < 64 is zero-padded to 64 (K' then carries 64 - D extra eigenvalues
// noise_c, which the formulas above absorb with D := 64).
#include "dkt_mfma_tiles.h"

namespace {

using namespace dkt_mfma;

constexpr int LR_DP = DKT_LOWRANK_DP;          // 64: feature dimension of the D x D problem (include/dkt_abi.h)
constexpr float LR_LOG_2PI = 1.8378770664093454836f;
constexpr int LR_U = 8;                         // K steps (of 4 rows) whose loads a wave of the Gram kernel keeps in flight

__device__ __forceinline__ f32x4 mfma4(const float a, const float b, const f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------------------------------------------------------------------------
// A[b] = Zp^T Zp (64 x 64, row-major, exactly symmetric), P[b, c, :] = Zp^T (y_c - m_c).  NCT = class tiles of 16.
template <int NCT>
__global__ __launch_bounds__(64) void lowrank_gram_kernel(const float* __restrict__ Z, const float* __restrict__ Y, const long y_bstride,
                                                          const float* __restrict__ mean, float* __restrict__ A, float* __restrict__ P,
                                                          const int C, const int N, const int D) {
    const int b = blockIdx.x, lane = threadIdx.x, kk = lane >> 4, m = lane & 15;
    const brsrc Zr = mk_rsrc(Z + (size_t)b * N * D, (unsigned)((size_t)N * D * 4));
    const brsrc Yr = mk_rsrc(Y + (size_t)b * y_bstride, (unsigned)((size_t)C * N * 4));
    const bool col_ok = 4 * m < D;
    float mc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) mc[ct] = (16 * ct + m < C) ? mean[16 * ct + m] : 0.f;
    f32x4 acc[10], accp[NCT][4];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) accp[ct][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K steps of 4 rows, LR_U at a time: the loads of the next chunk are issued before the products of the current one (a wave keeps LR_U 1-KB row groups
    // of Z in flight).  At cfg1 (27 K steps) this measures the same as one load in flight at seven waves per SIMD -- 0.097 ms per 8192 episodes either way:
    // the kernel sits at ~40 % of the fp32 MFMA rate, not on memory latency -- and is kept for long episodes (N = 420: 105 K steps at three waves per SIMD).
    const int nks = (N + 3) >> 2;
    f32x4 va[LR_U];
    float ra[LR_U][NCT];
    auto load_chunk = [&](const int k0, f32x4 (&v)[LR_U], float (&r)[LR_U][NCT]) {
#pragma unroll
        for (int u = 0; u < LR_U; ++u) {
            const int row = 4 * (k0 + u) + kk;
            v[u] = bload4(Zr, (col_ok && row < N) ? (row * D + 4 * m) * 4 : OOB, 0);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                // unconditional load (an out-of-range offset returns 0) and no select on the value: written as `ok ? y - m : 0` hipcc branches around
                // the load and waits vmcnt(0) inside the branch, which serialises every Z load of the chunk behind it (measured: 0.134 vs 0.097 ms per
                // 8192 cfg1 episodes).  Rows beyond N multiply zero rows of Z, classes beyond C have mc = 0 and are never stored.
                const int n = 16 * ct + m;
                const int off = (n < C && row < N) ? (n * N + row) * 4 : OOB;
                r[u][ct] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Yr, off, 0, 0)) - mc[ct];
            }
        }
    };
    load_chunk(0, va, ra);
    for (int k0 = 0; k0 < nks; k0 += LR_U) {
        f32x4 vb[LR_U];
        float rb[LR_U][NCT];
        load_chunk(k0 + LR_U, vb, rb);                              // (beyond the episode: every offset is out of range, the loads return 0)
#pragma unroll
        for (int u = 0; u < LR_U; ++u) {
            const f32x4 v = va[u];
            int t = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int q2 = q; q2 < 4; ++q2, ++t) acc[t] = mfma4(v[q], v[q2], acc[t]);
            }
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) accp[ct][q] = mfma4(v[q], ra[u][ct], accp[ct][q]);
        }
#pragma unroll
        for (int u = 0; u < LR_U; ++u) {
            va[u] = vb[u];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) ra[u][ct] = rb[u][ct];
        }
    }
    // accumulator lane (g, c), register r: element [16 q + 4 g + r][16 q2 + c]
    const int g4 = 4 * kk, c = m;
    const brsrc Ar = mk_rsrc(A + (size_t)b * LR_DP * LR_DP, LR_DP * LR_DP * 4);
    int t = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int q2 = q; q2 < 4; ++q2, ++t) {
            bstore4(Ar, acc[t], ((16 * q2 + c) * LR_DP + 16 * q + g4) * 4, 0);            // the mirrored block (rows c): one 16-byte store
            if (q2 > q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bstore1(Ar, acc[t][r], ((16 * q + g4 + r) * LR_DP + 16 * q2 + c) * 4, 0);
            }
        }
    }
    const brsrc Pr = mk_rsrc(P + (size_t)b * C * LR_DP, (unsigned)((size_t)C * LR_DP * 4));
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int n = 16 * ct + c;
#pragma unroll
        for (int q = 0; q < 4; ++q) bstore4(Pr, accp[ct][q], (n < C) ? (n * LR_DP + 16 * q + g4) * 4 : OOB, 0);
    }
}

// sum over the four row groups (lanes c, c + 16, c + 32, c + 48): every lane ends up with the total of its column
__device__ __forceinline__ float colsum(float v) {
    v += __shfl_xor(v, 16, DKT_WAVE);
    v += __shfl_xor(v, 32, DKT_WAVE);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Second pass over Z: S = Zp T^T (N x C), alpha = (r - sv S) / nz, V = cw sv (alpha - S), and the per-class scalars.
template <int NCT>
__global__ __launch_bounds__(64) void lowrank_finish_kernel(const float* __restrict__ Z, const float* __restrict__ Y, const long y_bstride,
                                                            const float* __restrict__ sv, const float* __restrict__ mean,
                                                            const float* __restrict__ noise, const float* __restrict__ cls_weight,
                                                            const float* __restrict__ T, const float* __restrict__ logp_d,
                                                            const float* __restrict__ dnoise_d, const float* __restrict__ jitter,
                                                            const float* __restrict__ pre_jitter, float* __restrict__ jitter_total,
                                                            float* __restrict__ obj,
                                                            float* __restrict__ logp, float* __restrict__ alpha, float* __restrict__ V,
                                                            float* __restrict__ dsv, float* __restrict__ dmean, float* __restrict__ dnoise,
                                                            const int C, const int N, const int D) {
    const int b = blockIdx.x, lane = threadIdx.x, kk = lane >> 4, m = lane & 15;
    const brsrc Zr = mk_rsrc(Z + (size_t)b * N * D, (unsigned)((size_t)N * D * 4));
    const brsrc Yr = mk_rsrc(Y + (size_t)b * y_bstride, (unsigned)((size_t)C * N * 4));
    const brsrc Tr = mk_rsrc(T + (size_t)b * C * LR_DP, (unsigned)((size_t)C * LR_DP * 4));
    const brsrc Alr = mk_rsrc(alpha + (size_t)b * C * N, (unsigned)((size_t)C * N * 4));
    const brsrc Vr = mk_rsrc(V + (size_t)b * C * N, (unsigned)((size_t)C * N * 4));
    // b operand of K step (j, q): t[class 16 ct + m][d' = 16 q + 4 j + kk]
    float tb[NCT][4][4], svc[NCT], mc[NCT], nz[NCT], cws[NCT], tt[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int n = 16 * ct + m;
        const bool ok = n < C;
        svc[ct] = ok ? sv[n] : 1.f;
        mc[ct] = ok ? mean[n] : 0.f;
        nz[ct] = ok ? noise[n] + jitter[(size_t)b * C + n] : 1.f;
        cws[ct] = ok ? (cls_weight ? cls_weight[n] : 1.f) * svc[ct] : 0.f;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Tr, ok ? (n * LR_DP + 16 * q + 4 * j + kk) * 4 : OOB, 0, 0));
                tb[ct][j][q] = v;
                s2 = fmaf(v, v, s2);
            }
        tt[ct] = colsum(s2);                                   // |t_c|^2
    }
    const int g4 = 4 * kk, c = m;
    float sa[NCT], saa[NCT], srr[NCT], srs[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) sa[ct] = saa[ct] = srr[ct] = srs[ct] = 0.f;
    const int nrt = (N + 15) >> 4;
    auto load_rows = [&](const int rt, f32x4 (&z4)[4]) {       // a operand: row 16 rt + m of this lane, columns 16 j + 4 kk .. + 3
        const int arow = 16 * rt + m;
#pragma unroll
        for (int j = 0; j < 4; ++j) z4[j] = bload4(Zr, (arow < N && 16 * j + 4 * kk < D) ? (arow * D + 16 * j + 4 * kk) * 4 : OOB, 0);
    };
    f32x4 za[4];
    load_rows(0, za);
    for (int rt = 0; rt < nrt; ++rt) {
        f32x4 zn[4];
        load_rows(rt + 1, zn);                                 // the next row tile's loads fly during this tile's products and stores
        float yv[NCT][4];
        if (16 * rt + 16 <= N) {                               // (uniform) a row tile inside the episode: the lane's four targets are 16 consecutive bytes
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int n = 16 * ct + c;
                const f32x4 y4 = bload4(Yr, (n < C) ? (n * N + 16 * rt + g4) * 4 : OOB, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) yv[ct][r] = y4[r];
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 16 * ct + c, row = 16 * rt + g4 + r;
                    yv[ct][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Yr, (n < C && row < N) ? (n * N + row) * 4 : OOB, 0, 0));
                }
        }
        f32x4 S[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x4 sj[4];                                       // four independent accumulator chains (a single one is 16 dependent MFMAs), summed in a fixed order
#pragma unroll
            for (int j = 0; j < 4; ++j) sj[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) sj[j] = mfma4(za[j][q], tb[ct][j][q], sj[j]);
            S[ct] = (sj[0] + sj[1]) + (sj[2] + sj[3]);
        }
        // accumulator lane (g, c), register r: S[row 16 rt + 4 g + r][class 16 ct + c] -- a lane's four registers are four CONSECUTIVE rows of one class: a row tile
        // inside the episode leaves as one 16-byte store per lane, class tile and output (round 6; as 4-byte stores -- 64 isolated dwords per instruction, 16 instructions
        // per row tile at 20 classes -- this kernel sat at 0.26 of HBM at the 20-way shape); the episode's last, partial tile keeps the masked 4-byte stores
        const bool whole_tile = 16 * rt + 16 <= N;                 // uniform
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int n = 16 * ct + c;
            f32x4 al4, v4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rt + g4 + r;
                const bool ok = n < C && row < N;
                const float rr = ok ? yv[ct][r] - mc[ct] : 0.f;
                const float s = ok ? S[ct][r] : 0.f;
                const float al = (rr - svc[ct] * s) / nz[ct];
                al4[r] = al;
                v4[r] = cws[ct] * (al - s);
                sa[ct] += al;
                saa[ct] = fmaf(al, al, saa[ct]);
                srr[ct] = fmaf(rr, rr, srr[ct]);
                srs[ct] = fmaf(rr, s, srs[ct]);
            }
            if (whole_tile) {
                const int off = (n < C) ? (n * N + 16 * rt + g4) * 4 : OOB;
                bstore4(Alr, al4, off, 0);
                bstore4(Vr, v4, off, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * rt + g4 + r;
                    const int off = (n < C && row < N) ? (n * N + row) * 4 : OOB;
                    bstore1(Alr, al4[r], off, 0);
                    bstore1(Vr, v4[r], off, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) za[j] = zn[j];
    }
    float objp = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const float a1 = colsum(sa[ct]), a2 = colsum(saa[ct]), r2 = colsum(srr[ct]), rs = colsum(srs[ct]);
        const int n = 16 * ct + c;
        if (kk == 0 && n < C) {
            const size_t bc = (size_t)b * C + n;
            const float s = svc[ct], z = nz[ct];
            const float lpd = logp_d[bc];                                             // NaN for a failed matrix: everything below is poisoned with it
            const float logdet_d = -2.0f * lpd - rs - (float)LR_DP * LR_LOG_2PI;       // log det K'_c   (rs = p^T K'^-1 p)
            const float quad = (r2 - s * rs) / z;
            const float trk_d = tt[ct] - 2.0f * dnoise_d[bc];                          // tr K'_c^-1
            const float trk = (float)(N - LR_DP) / z + trk_d;
            const float lp = -0.5f * quad - 0.5f * ((float)(N - LR_DP) * logf(z) + logdet_d) - (float)N * DKT_HALF_LOG_2PI;
            logp[bc] = lp;
            objp += (cls_weight ? cls_weight[n] : 1.f) * lp;
            if (jitter_total) jitter_total[bc] = jitter[bc] + (pre_jitter ? pre_jitter[n] : 0.f);
            dmean[bc] = a1;
            dnoise[bc] = 0.5f * (a2 - trk);
            dsv[bc] = 0.5f * (tt[ct] - ((float)LR_DP - z * trk_d) / s);
        }
    }
    // obj[b] = sum_c cls_weight_c logp[b, c]: the classes sit on lanes 0 .. 15 (row group 0) of each class tile; fixed order
    if (obj) {
        float o = (kk == 0) ? objp : 0.f;
#pragma unroll
        for (int sft = 8; sft > 0; sft >>= 1) o += __shfl_xor(o, sft, DKT_WAVE);
        if (lane == 0) obj[b] = o;
    }
}

// Which rung of psd_safe_cholesky's ladder (0, j0, 10 j0, ...: TOTAL jitter) the N x N matrix K_c = sv_c Z Z^T + noise_c I needs before it is numerically
// positive definite at all.  With D < N that matrix has N - D eigenvalues equal to noise_c, so its fp32 Cholesky fails -- and the reference retries -- when
// the noise floor drowns in the rounding of the diagonal: noise_c + jitter < 2^-22 max_i K_ii = 2^-22 (sv_c zmax2 + noise_c).  The D x D models never see that
// rank deficiency (sv_c Z^T Z is positive definite on its own), so the rung is chosen here, per class; the D x D call's own ladder runs on top of it.
// NaN where no rung clears the floor: the class then fails in the D x D call (info != 0, NaN outputs).
__global__ void lowrank_noise_floor_kernel(const float* __restrict__ sv, const float* __restrict__ noise, const float* __restrict__ zmax2, const float jitter0,
                                           const int max_tries, float* __restrict__ noise_eff, float* __restrict__ pre_jitter, const int C) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= C) return;
    const float nz = noise[n], zm = zmax2 ? zmax2[0] : 1.0f;
    const float thresh = (sv[n] * zm + nz) * 2.384185791015625e-07f;          // 2^-22
    float pre = __int_as_float(0x7fc00000), jit = 0.f;
    for (int attempt = 0; attempt <= max_tries; ++attempt) {
        if (attempt == 1) jit = jitter0;
        if (attempt > 1) jit *= 10.f;
        const float lifted = nz + jit;
        if (lifted >= thresh && lifted > 0.f) { pre = jit; break; }
    }
    pre_jitter[n] = pre;
    noise_eff[n] = nz + pre;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// dZ = g_b (V T^T + 2 Z W') in the permuted feature order, written back as full rows.  NKC = K steps of 4 classes.
template <int NKC>
__global__ __launch_bounds__(64) void lowrank_bwd_kernel(const float* __restrict__ Z, const float* __restrict__ V, const float* __restrict__ T,
                                                         const float* __restrict__ Wd, const float* __restrict__ ep_scale, float* __restrict__ dZ,
                                                         const int C, const int N, const int D) {
    const int b = blockIdx.x, lane = threadIdx.x, kk = lane >> 4, m = lane & 15;
    const brsrc Zr = mk_rsrc(Z + (size_t)b * N * D, (unsigned)((size_t)N * D * 4));
    const brsrc dZr = mk_rsrc(dZ + (size_t)b * N * D, (unsigned)((size_t)N * D * 4));
    const brsrc Vr = mk_rsrc(V + (size_t)b * C * N, (unsigned)((size_t)C * N * 4));
    const brsrc Tr = mk_rsrc(T + (size_t)b * C * LR_DP, (unsigned)((size_t)C * LR_DP * 4));
    const brsrc Wr = mk_rsrc(Wd + (size_t)b * LR_DP * LR_DP, LR_DP * LR_DP * 4);
    const float gb = ep_scale ? ep_scale[b] : 1.0f;
    // b operands, resident for the whole episode: 2 g W'[d' = 16 q + 4 j + kk][16 q2 + m] and g t[class 4 ks + kk][16 q2 + m]
    float wb[4][4][4], tb[NKC][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int q2 = 0; q2 < 4; ++q2)
                wb[j][q][q2] = 2.0f * gb * __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Wr, ((16 * q + 4 * j + kk) * LR_DP + 16 * q2 + m) * 4, 0, 0));
#pragma unroll
    for (int ks = 0; ks < NKC; ++ks)
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            const int n = 4 * ks + kk;
            tb[ks][q2] = gb * __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Tr, (n < C) ? (n * LR_DP + 16 * q2 + m) * 4 : OOB, 0, 0));
        }
    const int g4 = 4 * kk, c = m;
    const int nrt = (N + 15) >> 4;
    auto load_rows = [&](const int rt, f32x4 (&z4)[4], float (&v1)[NKC]) {
        const int arow = 16 * rt + m;
#pragma unroll
        for (int j = 0; j < 4; ++j) z4[j] = bload4(Zr, (arow < N && 16 * j + 4 * kk < D) ? (arow * D + 16 * j + 4 * kk) * 4 : OOB, 0);
#pragma unroll
        for (int ks = 0; ks < NKC; ++ks) {
            const int n = 4 * ks + kk;
            v1[ks] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(Vr, (n < C && arow < N) ? (n * N + arow) * 4 : OOB, 0, 0));
        }
    };
    f32x4 za[4];
    float va[NKC];
    load_rows(0, za, va);
    for (int rt = 0; rt < nrt; ++rt) {
        f32x4 zn[4];
        float vn[NKC];
        load_rows(rt + 1, zn, vn);                             // the next row tile's operands fly during this tile's 64 + 4 NKC products
        f32x4 out[4];
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            out[q2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKC; ++ks) out[q2] = mfma4(va[ks], tb[ks][q2], out[q2]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) out[q2] = mfma4(za[j][q], wb[j][q][q2], out[q2]);
        }
        // accumulator lane (g, c), register r: dZ'[row 16 rt + 4 g + r][d' = 16 q2 + c] = column 4 c + q2: the four tiles of a lane are 16 consecutive bytes
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * rt + g4 + r;
            const f32x4 o = {out[0][r], out[1][r], out[2][r], out[3][r]};
            bstore4(dZr, o, (row < N && 4 * c < D) ? (row * D + 4 * c) * 4 : OOB, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) za[j] = zn[j];
#pragma unroll
        for (int ks = 0; ks < NKC; ++ks) va[ks] = vn[ks];
    }
}

bool lr_shape_ok(int B, int C, int N, int D) {
    return B > 0 && C > 0 && C <= 32 && N > 0 && D > 0 && D <= LR_DP && (D & 3) == 0 && (size_t)N * D * 4 < 0x7fffff00u && (size_t)C * N * 4 < 0x7fffff00u;
}

}  // namespace

extern "C" int dkt_lowrank_supported(int C, int N, int D) { return lr_shape_ok(1, C, N, D) ? 1 : 0; }

extern "C" int dkt_lowrank_gram_f32(const float* Z, const float* Y, long y_bstride, const float* mean, float* A, float* P,
                                    int B, int C, int N, int D, void* stream) {
    if (!Z || !Y || !mean || !A || !P || y_bstride < 0) return DKT_ERR_BAD_ARG;
    if (B <= 0 || C <= 0 || N <= 0 || D <= 0 || (D & 3) != 0) return DKT_ERR_BAD_ARG;
    if (!lr_shape_ok(B, C, N, D)) return DKT_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
    if (C <= 16) hipLaunchKernelGGL((lowrank_gram_kernel<1>), dim3(B), dim3(64), dkt_lds_pad("DKT_PAD_LR_GRAM"), st, Z, Y, y_bstride, mean, A, P, C, N, D);
    else hipLaunchKernelGGL((lowrank_gram_kernel<2>), dim3(B), dim3(64), dkt_lds_pad("DKT_PAD_LR_GRAM"), st, Z, Y, y_bstride, mean, A, P, C, N, D);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_lowrank_noise_floor_f32(const float* sv, const float* noise, const float* zmax2, float jitter0, int max_tries, float* noise_eff,
                                           float* pre_jitter, int C, void* stream) {
    if (!sv || !noise || !noise_eff || !pre_jitter || C <= 0 || max_tries < 0 || max_tries > 8) return DKT_ERR_BAD_ARG;
    hipLaunchKernelGGL(lowrank_noise_floor_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sv, noise, zmax2, jitter0, max_tries, noise_eff, pre_jitter, C);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_lowrank_finish_f32(const float* Z, const float* Y, long y_bstride, const float* sv, const float* mean, const float* noise,
                                      const float* cls_weight, const float* T, const float* logp_d, const float* dnoise_d, const float* jitter_used,
                                      const float* pre_jitter, float* jitter_total, float* obj,
                                      float* logp, float* alpha, float* V, float* dsv, float* dmean, float* dnoise,
                                      int B, int C, int N, int D, void* stream) {
    if (!Z || !Y || !sv || !mean || !noise || !T || !logp_d || !dnoise_d || !jitter_used || !logp || !alpha || !V || !dsv || !dmean || !dnoise || y_bstride < 0)
        return DKT_ERR_BAD_ARG;
    if (B <= 0 || C <= 0 || N <= 0 || D <= 0 || (D & 3) != 0) return DKT_ERR_BAD_ARG;
    if (!lr_shape_ok(B, C, N, D)) return DKT_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
    if (C <= 16)
        hipLaunchKernelGGL((lowrank_finish_kernel<1>), dim3(B), dim3(64), dkt_lds_pad("DKT_PAD_LR_FIN"), st, Z, Y, y_bstride, sv, mean, noise, cls_weight, T, logp_d, dnoise_d, jitter_used,
                           pre_jitter, jitter_total, obj, logp, alpha, V, dsv, dmean, dnoise, C, N, D);
    else
        hipLaunchKernelGGL((lowrank_finish_kernel<2>), dim3(B), dim3(64), dkt_lds_pad("DKT_PAD_LR_FIN"), st, Z, Y, y_bstride, sv, mean, noise, cls_weight, T, logp_d, dnoise_d, jitter_used,
                           pre_jitter, jitter_total, obj, logp, alpha, V, dsv, dmean, dnoise, C, N, D);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_lowrank_bwd_f32(const float* Z, const float* V, const float* T, const float* Wd, const float* ep_scale, float* dZ,
                                   int B, int C, int N, int D, void* stream) {
    if (!Z || !V || !T || !Wd || !dZ) return DKT_ERR_BAD_ARG;
    if (B <= 0 || C <= 0 || N <= 0 || D <= 0 || (D & 3) != 0) return DKT_ERR_BAD_ARG;
    if (!lr_shape_ok(B, C, N, D)) return DKT_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
    const int nkc = (C + 3) / 4;
#define DKT_LR_BWD(K) hipLaunchKernelGGL((lowrank_bwd_kernel<K>), dim3(B), dim3(64), dkt_lds_pad("DKT_PAD_LR_BWD"), st, Z, V, T, Wd, ep_scale, dZ, C, N, D)
    if (nkc <= 2) DKT_LR_BWD(2);
    else if (nkc <= 4) DKT_LR_BWD(4);
    else if (nkc <= 5) DKT_LR_BWD(5);
    else DKT_LR_BWD(8);
#undef DKT_LR_BWD
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
