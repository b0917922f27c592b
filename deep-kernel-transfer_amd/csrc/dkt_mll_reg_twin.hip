// dkt_mll_reg_twin.hip -- register-resident exact-GP marginal likelihood for N + 1 <= 128 (the few-shot
// shapes: N = 105, 85, 25, 19, 5).
//
// One 256-thread workgroup per episode, classes in sequence.  The (N+1) x N working matrix of
// dkt_mll.hip (L below the diagonal, U = L^-T above it, w = L^-1 r in row N) never touches LDS: it is
// distributed 2-D cyclically over the 16 x 16 thread grid,
//       thread (ty, tx) owns  Mw[ty + 16*pi][tx + 16*ji],  pi, ji in [0, NT),  NT = ceil((N+1)/16),
// i.e. NT*NT registers per lane (49 for N = 105), rows held in pairs (pi = 2m, 2m+1) so the rank-1 update runs on
// v_pk_fma_f32.  A sweep step costs ONE barrier and ONE LDS round trip: the 16 owners of column k publish it (raw) to a
// double-buffered LDS vector; every thread reads the pivot and its NT row / NT column entries, takes v_rcp_f32, updates
// block column KQ first, the owners of column k+1 publish it at once, and the remaining block columns are updated off
// the critical path.  KQ = k / 16 is a template parameter, so every register index is static and blocks that are
// structurally untouched (ji < KQ, or KQ < pi < ji) cost nothing.  Columns stay unscaled during the sweep; pivots,
// column scales 1 / L_kk, log det and the first non-positive pivot are all read off the diagonal slots afterwards.
//
// Gradient: K^-1 = U U^T is a Gram matrix of the rows of U.  Per class, U goes through LDS one 16-column chunk at a
// time and each wave accumulates its tile rows of U U^T on v_mfma_f32_16x16x4_f32 (four per-wave instantiations, each
// accumulator first touched when its row block becomes non-zero); the epilogue forms coef_c (alpha alpha^T - U U^T)
// with alpha from LDS and adds it into W[b] in memory, class after class (same lane, same word: deterministic).
// The per-class hyper gradients need only scalars:  tr K^-1 = |U|_F^2,  alpha.alpha,  1.alpha,  r.alpha:
//     dnoise = 0.5 (alpha.alpha - tr K^-1)
//     dsv    = 0.5 ((r.alpha - N) - (noise + jitter) (alpha.alpha - tr K^-1)) / sv      [sv E = K - (noise+jitter) I]
//
// Replaces the same reference lines as dkt_mll.hip (methods/DKT.py:161-163,177,187,252-254,265,330;
// methods/DKT_regression.py:53-56,92).
//
// Round 4: the round-1 default, since round 2 only the validation twin of the MFMA wave-per-matrix kernels -- built into libdkt_diag.so (tests / tools), no longer
// into the product library: dkt_diag_mll_reg_f32 below takes the arguments of dkt_mll_f32 (include/dkt_abi.h) without the workspace.
#include "dkt_reg_sweep.h"
#include "../../include/dkt_abi.h"

namespace {

template <int NT, bool WANT_GRAD, bool WANT_CHOL>
#ifndef DKT_REG_MINW
#define DKT_REG_MINW 4
#endif
__global__ __launch_bounds__(256, ((NT <= 7 && !(WANT_GRAD && WANT_CHOL)) ? DKT_REG_MINW : 2)) void mll_reg_kernel(MllArgs a) {
    constexpr int NP = 16 * NT;
    __shared__ float colbuf[4 * NP];           // two double-buffered pivot-column pairs
    __shared__ float wv[NP];
    __shared__ float dgv[NP];
    __shared__ __attribute__((aligned(16))) float alv[NP];
    __shared__ float red[20];
    __shared__ __attribute__((aligned(16))) float ubuf[2 * NP * ULD];

    const int b = blockIdx.x, tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int lane = tid & 63, r16 = lane & 15, q = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave id in an SGPR: scalar switch
    const int N = a.N, C = a.C;
    const int tyN = N - 16 * (NT - 1);                  // row N lives at pi = NT-1, ty = tyN; column N at ji = NT-1, tx = tyN
    const bool lower_eq = ty >= tx, upper_eq = tx >= ty;
    const bool row_ok = ty < tyN, is_w = ty == tyN;      // last block row: p < N / p == N
    const bool col_ok = tx < tyN, is_acol = tx == tyN;    // last block column: j < N / j == N
    const brsrc Er = make_rsrc(a.E + (size_t)b * N * N, N * N * 4);
    const int vo_form = (ty * N + tx) * 4;               // per-lane part of element (ty + 16 pi, tx + 16 ji)
    const int vo_rc = (4 * q * N + r16) * 4, vo_cr = (r16 * N + 4 * q) * 4;   // MFMA-layout (row 4q, col r16) and its mirror
    constexpr bool want_grad = WANT_GRAD;
    constexpr bool want_chol = WANT_CHOL;

    Masks masks;
    masks.row_ok = row_ok; masks.col_ok = col_ok; masks.is_acol = is_acol;
    RegCtx<NT> ctx;
    ctx.colbuf = colbuf; ctx.N = N; ctx.tx = tx; ctx.ty = ty; ctx.tid = tid; ctx.col_ok = col_ok;

    bool poisoned = false;

    for (int c = 0; c < C; ++c) {
        const float svc = a.sv[c], mc = a.mean[c], nzc = a.noise[c];
        const brsrc yr = make_rsrc(a.Y + (size_t)b * a.y_bstride + (size_t)c * N, N * 4);
        f32x2 A2[(NT + 1) / 2][NT];
        float rinvcol[NT];
        float log2part = 0.f;                            // this thread's share of sum_k log2 d_k (ty == 0 lanes)
        int fail_at = 0;
        float jit = 0.f;
        for (int attempt = 0; attempt <= a.max_tries; ++attempt) {
            jit = 0.f;
            if (attempt > 0) {
                jit = a.jitter0;
                for (int i = 1; i < attempt; ++i) jit *= 10.f;
            }
#pragma unroll
            for (int pi = 0; pi < NT; ++pi) {
#pragma unroll
                for (int ji = 0; ji < NT; ++ji) {
                    // static block structure; only the last block row / column need per-thread masks
                    float v = 0.f;
                    if (pi >= ji) {
                        bool ld = true;                                    // lower-triangle element of K?
                        if (pi == ji) ld = lower_eq;
                        if (pi == NT - 1) ld = ld && row_ok;
                        if (ji == NT - 1) ld = ld && col_ok;
                        {
                            float x = svc * bload(Er, vo_form, (16 * pi * N + 16 * ji) * 4);   // out-of-range reads return 0
                            if (pi == ji && tx == ty) x += nzc + jit;
                            v = ld ? x : 0.f;
                        }
                        if (pi == NT - 1) {
                            bool lw = is_w;
                            if (ji == NT - 1) lw = lw && col_ok;
                            const float yv = bload(yr, tx * 4, 16 * ji * 4) - mc;
                            v = lw ? yv : v;
                        }
                    }
                    AE(pi, ji) = v;
                }
            }
            __syncthreads();          // previous users of colbuf are done
            sweep_all<NT, 0>(A2, ctx);
            __builtin_amdgcn_s_setprio(0);
            // ---- pivots: d_j sits raw in the diagonal slot (j, j).  Column scales, log det, first bad pivot ----
            if (ty == tx) {
#pragma unroll
                for (int ji = 0; ji < NT; ++ji) dgv[tx + 16 * ji] = AE(ji, ji);
            }
            __syncthreads();
            int bad = 0x7fffffff;
            log2part = 0.f;
#pragma unroll
            for (int ji = NT - 1; ji >= 0; --ji) {
                const float dj = dgv[tx + 16 * ji];
                const bool valid = (ji < NT - 1) || col_ok;
                rinvcol[ji] = valid ? __builtin_amdgcn_rsqf(dj) : 1.0f;
                log2part += valid ? __builtin_amdgcn_logf(dj) : 0.f;          // v_log_f32 = log2
                bad = (valid && !(dj > 0.f)) ? tx + 16 * ji + 1 : bad;        // descending ji: the smallest index wins
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) bad = min(bad, __shfl_xor(bad, o, DKT_WAVE));
            bad = __builtin_amdgcn_readfirstlane(bad);       // every 16-lane row of every wave computed the same value
            fail_at = (bad == 0x7fffffff) ? 0 : bad;
            if (fail_at == 0) break;
        }
        const size_t bc = (size_t)b * C + c;
        if (fail_at != 0) {
            const float qnan = __int_as_float(0x7fc00000);
            if (tid == 0) {
                a.logp[bc] = qnan;
                a.jitter_used[bc] = jit;
                a.info[bc] = fail_at;
                if (want_grad) { a.dsv[bc] = qnan; a.dmean[bc] = qnan; a.dnoise[bc] = qnan; }
            }
            for (int i = tid; i < N; i += 256) a.alpha[bc * N + i] = qnan;
            if constexpr (want_chol)
                for (int idx = tid; idx < N * N; idx += 256) a.L[bc * N * N + idx] = qnan;
            poisoned = true;
            __syncthreads();
            continue;
        }
        // ---- apply the lazy column scaling: column k gets 1 / L_kk, the diagonal slot becomes U_kk = 1 / L_kk ----
#pragma unroll
        for (int pi = 0; pi < NT; ++pi) {
#pragma unroll
            for (int ji = 0; ji < NT; ++ji) {
                AE(pi, ji) *= rinvcol[ji];
                if (pi == ji) {
                    bool dg = tx == ty;
                    if (pi == NT - 1) dg = dg && col_ok;           // the (N, N) corner and beyond are padding, not pivots
                    AE(pi, ji) = dg ? rinvcol[ji] : AE(pi, ji);
                }
            }
        }
        // ---- w row -> LDS; scalars ----
        if (ty == tyN) {
#pragma unroll
            for (int ji = 0; ji < NT; ++ji) wv[tx + 16 * ji] = AE(NT - 1, ji);
        }
        __syncthreads();
        float wj[NT];
#pragma unroll
        for (int ji = 0; ji < NT; ++ji) wj[ji] = wv[tx + 16 * ji];
        float v5[5] = {0.f, (ty == 0) ? log2part : 0.f, 0.f, 0.f, 0.f};     // quad, sum log2 d, sum alpha, sum alpha^2, |U|_F^2
        if (ty == tyN) {
#pragma unroll
            for (int ji = 0; ji < NT; ++ji) v5[0] += wj[ji] * wj[ji];
        }
#pragma unroll
        for (int pi = 0; pi < NT; ++pi) {
            const int p = ty + 16 * pi;
            float s = 0.f, u2 = 0.f;
#pragma unroll
            for (int ji = pi; ji < NT; ++ji) {
                float u = AE(pi, ji);
                if (ji == pi) u = upper_eq ? u : 0.f;              // strictly-lower entries of the diagonal block are L
                if (ji == NT - 1) u = col_ok ? u : 0.f;            // padding / alpha slot
                s += u * wj[ji];
                u2 += u * u;
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, DKT_WAVE);
            if (pi == NT - 1) { s = row_ok ? s : 0.f; u2 = row_ok ? u2 : 0.f; }
            v5[4] += u2;
            if (tx == 0) {
                alv[p] = s;                                         // alpha_p for the gradient product (column N of [U | alpha])
                v5[2] += s;
                v5[3] += s * s;
                if (pi < NT - 1 || row_ok) a.alpha[bc * N + p] = s;
            }
        }
        block_sum5(v5, red);
        const float quad = v5[0], logdet_half = 0.34657359027997264f * v5[1], asum = v5[2], a2 = v5[3], trk = v5[4];
        if (tid == 0) {
            a.logp[bc] = -0.5f * quad - logdet_half - (float)N * DKT_HALF_LOG_2PI;
            a.jitter_used[bc] = jit;
            a.info[bc] = 0;
            if (want_grad) {
                const float nz_eff = nzc + jit;
                a.dmean[bc] = asum;
                a.dnoise[bc] = 0.5f * (a2 - trk);
                a.dsv[bc] = 0.5f * ((quad - (float)N) - nz_eff * (a2 - trk)) / svc;
            }
        }
        if constexpr (want_chol) {
            float* Lb = a.L + bc * N * N;
#pragma unroll
            for (int pi = 0; pi < NT; ++pi) {
#pragma unroll
                for (int ji = 0; ji < NT; ++ji) {
                    const int p = ty + 16 * pi, j = tx + 16 * ji;
                    bool ok = true;
                    if (pi == NT - 1) ok = ok && row_ok;
                    if (ji == NT - 1) ok = ok && col_ok;
                    if (ok) {
                        float v = 0.f;
                        if (pi > ji) v = AE(pi, ji);
                        else if (pi == ji) v = (ty > tx) ? AE(pi, ji) : ((ty == tx) ? 1.0f / rinvcol[ji] : 0.f);
                        Lb[p * N + j] = v;
                    }
                }
            }
        }
        if constexpr (want_grad) {
            const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
            const float coef = 0.5f * cw * svc;
            // W += coef (alpha alpha^T - U U^T): U U^T on MFMA, the rank-1 alpha term in the epilogue (alpha from LDS)
            // The class's contribution is accumulated into W[b] in memory (same lanes, same words, class after class:
            // deterministic, no atomics) so that no accumulator registers stay live across the next class's sweep.
            const brsrc Wr = make_rsrc(a.W + (size_t)b * N * N, N * N * 4);
            const bool first = (c == 0), last = (c == C - 1);
            switch (wave_u) {
                case 0: w_product_wave<NT, 0>(A2, ubuf, tx, ty, r16, q, masks, Wr, N, tyN, vo_rc, vo_cr, first, last, coef, alv); break;
                case 1: w_product_wave<NT, 1>(A2, ubuf, tx, ty, r16, q, masks, Wr, N, tyN, vo_rc, vo_cr, first, last, coef, alv); break;
                case 2: w_product_wave<NT, 2>(A2, ubuf, tx, ty, r16, q, masks, Wr, N, tyN, vo_rc, vo_cr, first, last, coef, alv); break;
                default: w_product_wave<NT, 3>(A2, ubuf, tx, ty, r16, q, masks, Wr, N, tyN, vo_rc, vo_cr, first, last, coef, alv); break;
            }
        }
        __syncthreads();
    }

    if constexpr (want_grad) {
        if (poisoned) {
            float* Wb = a.W + (size_t)b * N * N;
            const float qnan = __int_as_float(0x7fc00000);
            for (int idx = tid; idx < N * N; idx += 256) Wb[idx] = qnan;
        }
    }
}

template <int NT>
void launch_reg(const MllArgs& a, hipStream_t st) {
    const bool g = (a.flags & DKT_MLL_WANT_GRAD) != 0, c = (a.flags & DKT_MLL_WANT_CHOL) != 0;
    if (g && c) hipLaunchKernelGGL((mll_reg_kernel<NT, true, true>), dim3(a.B), dim3(256), 0, st, a);
    else if (g) hipLaunchKernelGGL((mll_reg_kernel<NT, true, false>), dim3(a.B), dim3(256), 0, st, a);
    else if (c) hipLaunchKernelGGL((mll_reg_kernel<NT, false, true>), dim3(a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((mll_reg_kernel<NT, false, false>), dim3(a.B), dim3(256), 0, st, a);
}

}  // namespace

bool dkt_mll_reg_launch(const MllArgs& a, hipStream_t st) {
    const int nt = (a.N + 1 + 15) / 16;
    switch (nt) {
        case 1: launch_reg<1>(a, st); return true;
        case 2: launch_reg<2>(a, st); return true;
        case 3: launch_reg<3>(a, st); return true;
        case 4: launch_reg<4>(a, st); return true;
        case 5: launch_reg<5>(a, st); return true;
        case 6: launch_reg<6>(a, st); return true;
        case 7: launch_reg<7>(a, st); return true;
        case 8: launch_reg<8>(a, st); return true;
        default: return false;
    }
}

extern "C" int dkt_diag_mll_reg_f32(const float* E, const float* Y, long y_bstride, const float* sv, const float* mean, const float* noise, int B, int C, int N,
                                    float jitter0, int max_tries, unsigned flags, const float* cls_weight, float* logp, float* alpha, float* L, float* W, float* dsv,
                                    float* dmean, float* dnoise, float* jitter_used, int32_t* info, void* stream) {
    if (!E || !Y || !sv || !mean || !noise || !logp || !alpha || !jitter_used || !info) return DKT_ERR_BAD_ARG;
    if (B <= 0 || C <= 0 || N <= 0 || max_tries < 0 || max_tries > 8 || y_bstride < 0) return DKT_ERR_BAD_ARG;
    if ((flags & DKT_MLL_WANT_GRAD) && (!W || !dsv || !dmean || !dnoise)) return DKT_ERR_BAD_ARG;
    if ((flags & DKT_MLL_WANT_CHOL) && !L) return DKT_ERR_BAD_ARG;
    if (flags & ~(DKT_MLL_WANT_GRAD | DKT_MLL_WANT_CHOL)) return DKT_ERR_BAD_ARG;
    if (N + 1 > 128) return DKT_ERR_TOO_LARGE;
    MllArgs a;
    a.E = E; a.Y = Y; a.y_bstride = y_bstride; a.sv = sv; a.mean = mean; a.noise = noise;
    a.cls_weight = cls_weight; a.logp = logp; a.alpha = alpha; a.L = L; a.W = W; a.dsv = dsv;
    a.dmean = dmean; a.dnoise = dnoise; a.jitter_used = jitter_used; a.info = info;
    a.ws = nullptr; a.only_failed = nullptr; a.b0 = 0; a.B = B; a.C = C; a.N = N; a.LD = N | 1;                // (the leading dimension of the generic kernel; unused here)
    a.jitter0 = jitter0; a.max_tries = max_tries; a.flags = flags; a.p2_guard = 1;
    if (!dkt_mll_reg_launch(a, (hipStream_t)stream)) return DKT_ERR_TOO_LARGE;
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
