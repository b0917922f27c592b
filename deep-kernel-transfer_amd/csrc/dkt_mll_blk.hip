// dkt_mll_blk.hip -- BLOCKED right-looking exact-GP marginal likelihood for N + 1 <= 128:
// the 16-wide panel is factored with the register sweep, the trailing matrix is updated by MFMA.
//
// One 256-thread workgroup per episode, classes in sequence (as dkt_mll_reg.hip).  The (N+1) x N working
// matrix (L below the diagonal, U = L^-T above it, w = L^-1 r in row N; see dkt_mll.hip) lives in
// registers as 16x16 TILES IN MFMA ACCUMULATOR LAYOUT (lane (r16, q) holds rows 4q..4q+3 of column r16).
// Wave w owns the tile ROWS I = w and I = w + 4 (all NT column blocks): T[ri][J], 2 NT tiles = 56 VGPRs at
// N = 105.  Every wave runs the SAME code; its row indices are wave-uniform scalars (readfirstlane), so the
// structural cases (I == J, I < J, ...) are scalar branches, not per-wave code variants.
// Block step J (NT = ceil((N+1)/16) of them):
//   1. panel sweep, <= 16 sequential column steps INSIDE the panel only: the lanes holding column k publish
//      it (one ds_write_b128 per owned row block), ONE barrier, every lane reads pivot + its 4 row factors
//      (ds_read_b128) + its column factor and applies <= 8 v_fma -- a few dozen instructions per wave per
//      step instead of ~130 in the unblocked register kernel;
//   2. the factored panel goes to LDS ([16 NT][16] floats, b128-fragment friendly);
//   3. every trailing tile (I, J'), J' > J, that the sweep would touch (U rows I <= J, L rows I >= J') gets
//      T -= P_I P_J'^T with four v_mfma_f32_16x16x4_f32 (diagonal-block A operand masked to its upper
//      triangle, diagonal L tiles masked to their lower triangle).
// The result is the same matrix as the unblocked sweep up to fp32 summation order.
// alpha = U w and K^-1 = U U^T (training) reuse the panel buffer: U goes through LDS one 16-column chunk at
// a time; W = sum_c coef_c (alpha alpha^T - U U^T) accumulates over the classes in MFMA accumulators.
// Global accesses use buffer instructions (one per-lane VGPR offset + a scalar offset) so that no 64-bit
// per-element addresses exist for the compiler to hoist out of the class loop.
//
// Replaces the same reference lines as dkt_mll.hip (methods/DKT.py:161-163,177,187,252-254,265,330;
// methods/DKT_regression.py:53-56,92).
#include "dkt_mll.h"

namespace {

constexpr int PLD = 24;   // LDS row stride (floats) of the [16 NT][16] panel / chunk buffer: conflict-free b128 fragments

typedef __amdgpu_buffer_rsrc_t brsrc;
__device__ __forceinline__ brsrc make_rsrc(const void* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float bload(brsrc r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(brsrc r, float v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

template <int NT>
struct BlkCtx {
    int N, tyN, r16, q, tid;
    int wave;         // wave-uniform (SGPR)
    int vo_rc;        // byte offset of element (row 4q, col r16) of an [N][N] matrix: per-lane part of every address
    int vo_cr;        // byte offset of element (row r16, col 4q): per-lane part of the mirrored address
    bool col_ok;      // last block column: column 16 (NT-1) + r16 < N
};

constexpr int NRI = 2;                                   // tile rows per wave: I = wave + 4 ri
template <int NT> constexpr int acc_slots() { return NT + (NT > 4 ? NT - 4 : 0); }

// ---- form K_c (lower tiles), identity rows (zero), r row ------------------------------------------------
template <int NT>
__device__ __forceinline__ void blk_form(f32x4 (&T)[NRI][NT], brsrc Er, brsrc yr, float svc, float mc, float diagadd,
                                         const BlkCtx<NT>& c) {
#pragma unroll
    for (int ri = 0; ri < NRI; ++ri) {
        const int I = c.wave + 4 * ri;
#pragma unroll
        for (int J = 0; J < NT; ++J) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (I < NT && I >= J) {
                bool jok = true;
                if (J == NT - 1) jok = c.col_ok;
                const bool last = I == NT - 1, dgb = I == J;
                float yv = 0.f;
                if (last) yv = bload(yr, c.r16 * 4, 16 * J * 4) - mc;     // out-of-range columns read 0
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int pl = 4 * c.q + reg;                         // row inside the block
                    float x = svc * bload(Er, c.vo_rc, ((16 * I + reg) * c.N + 16 * J) * 4);
                    if (dgb && pl == c.r16) x += diagadd;
                    bool ld = jok;
                    if (dgb) ld = ld && (pl >= c.r16);
                    if (last) ld = ld && (pl < c.tyN);
                    x = ld ? x : 0.f;
                    if (last) x = (jok && pl == c.tyN) ? yv : x;
                    v[reg] = x;
                }
            }
            T[ri][J] = v;
        }
    }
}

// ---- one block column J of the blocked sweep ------------------------------------------------------------
// Returns 0 or (k+1) of the first non-positive pivot (block-uniform).
template <int NT, int J>
__device__ __forceinline__ int blk_block(f32x4 (&T)[NRI][NT], float* colbuf, float* pb, float& log2sum, const BlkCtx<NT>& c) {
    constexpr int NP = 16 * NT;
    const int kend = min(16, c.N - 16 * J);
    float rinv_mine = 1.0f;
    for (int kr = 0; kr < kend; ++kr) {
        float* cb = colbuf + (kr & 1) * NP;
        if (c.r16 == kr) {
#pragma unroll
            for (int ri = 0; ri < NRI; ++ri) {
                const int I = c.wave + 4 * ri;
                if (I < NT) *reinterpret_cast<f32x4*>(&cb[16 * I + 4 * c.q]) = T[ri][J];
            }
        }
        __syncthreads();
        const float d = cb[16 * J + kr];
        if (!(d > 0.f)) return 16 * J + kr + 1;
        const float rinv = __builtin_amdgcn_rsqf(d);
        log2sum += __builtin_amdgcn_logf(d);
        float cj = cb[16 * J + c.r16] * rinv;
        cj = (c.r16 > kr) ? cj : 0.f;                       // only columns j > k inside the panel
        if (J == NT - 1) cj = c.col_ok ? cj : 0.f;          // padding columns
#pragma unroll
        for (int ri = 0; ri < NRI; ++ri) {
            const int I = c.wave + 4 * ri;
            if (I < NT) {
                f32x4 cp = *reinterpret_cast<const f32x4*>(&cb[16 * I + 4 * c.q]) * rinv;
                if (I == J) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int pl = 4 * c.q + reg;
                        const float x = (pl == kr) ? rinv : cp[reg];              // row k itself: U_kk = 1 / L_kk
                        cp[reg] = ((pl <= kr) || (pl >= c.r16)) ? x : 0.f;        // p <= k (U row) or p >= j (L row)
                    }
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) T[ri][J][reg] = __builtin_fmaf(-cp[reg], cj, T[ri][J][reg]);
            }
        }
        rinv_mine = (c.r16 == kr) ? rinv : rinv_mine;
    }
    // lazy column scaling of the finished panel + copy to the LDS panel buffer
#pragma unroll
    for (int ri = 0; ri < NRI; ++ri) {
        const int I = c.wave + 4 * ri;
        if (I < NT) {
            const bool dgb = I == J;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float x = T[ri][J][reg] * rinv_mine;
                bool dg = dgb && ((4 * c.q + reg) == c.r16);
                if (J == NT - 1) dg = dg && c.col_ok;                             // the padding corner is not a pivot
                x = dg ? rinv_mine : x;
                T[ri][J][reg] = x;
                pb[(16 * I + 4 * c.q + reg) * PLD + c.r16] = x;
            }
        }
    }
    __syncthreads();
    // trailing tiles (I, JP), JP > J: T -= P_I P_JP^T where the sweep would have touched them
    if constexpr (J + 1 < NT) {
#pragma unroll
        for (int ri = 0; ri < NRI; ++ri) {
            const int I = c.wave + 4 * ri;
            if (I < NT) {
                f32x4 av = *reinterpret_cast<const f32x4*>(&pb[(16 * I + c.r16) * PLD + 4 * c.q]);
                if (I == J) {                                                     // diagonal block rows: U part only (k >= p)
#pragma unroll
                    for (int t = 0; t < 4; ++t) av[t] = (4 * c.q + t >= c.r16) ? av[t] : 0.f;
                }
                av = -av;
#pragma unroll
                for (int JP = J + 1; JP < NT; ++JP) {
                    if (I <= J || I >= JP) {
                        f32x4 bv = *reinterpret_cast<const f32x4*>(&pb[(16 * JP + c.r16) * PLD + 4 * c.q]);
                        if (JP == NT - 1) {                                       // panel rows >= N (w row, padding) are not columns
#pragma unroll
                            for (int t = 0; t < 4; ++t) bv[t] = c.col_ok ? bv[t] : 0.f;
                        }
                        if (I == JP) {                                            // diagonal L tile: only p >= j
                            f32x4 tmp = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int t = 0; t < 4; ++t) tmp = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[t], tmp, 0, 0, 0);
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) T[ri][JP][reg] += (4 * c.q + reg >= c.r16) ? tmp[reg] : 0.f;
                        } else {
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                T[ri][JP] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[t], T[ri][JP], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    return 0;
}

template <int NT, int J>
__device__ __forceinline__ int blk_all(f32x4 (&T)[NRI][NT], float* colbuf, float* pb, float& log2sum, const BlkCtx<NT>& c) {
    if constexpr (J < NT) {
        if (16 * J >= c.N) return 0;
        const int f = blk_block<NT, J>(T, colbuf, pb, log2sum, c);
        if (f) return f;
        return blk_all<NT, J + 1>(T, colbuf, pb, log2sum, c);
    } else {
        return 0;
    }
}

// value of tile (I, J) element as a U entry (upper incl. diagonal, rows / cols < N), else 0
template <int NT>
__device__ __forceinline__ float u_entry(float x, int I, int J, int pl, const BlkCtx<NT>& c) {
    if (I == J) x = (pl <= c.r16) ? x : 0.f;
    if (J == NT - 1) x = c.col_ok ? x : 0.f;
    if (I == NT - 1) x = (pl < c.tyN) ? x : 0.f;
    return x;
}

// W tile rows of this wave: RA = NT-1-wave, RB = wave - (8 - NT) (NT > 4), each valid when >= 0
template <int NT> __device__ __forceinline__ int w_row_a(int wave) { return NT - 1 - wave; }
template <int NT> __device__ __forceinline__ int w_row_b(int wave) { return NT > 4 ? wave - (8 - NT) : -1; }

// chunks of U (column block CH): alpha accumulation (threads tid < 16 (CH+1): row tid) + W product
template <int NT, int CH, bool WANT_GRAD>
__device__ __forceinline__ void blk_u_pass(const f32x4 (&T)[NRI][NT], f32x4* acc, float* pb, const float* wv, float& alpha_p,
                                           float ncoef, const BlkCtx<NT>& c) {
    if constexpr (CH < NT) {
#pragma unroll
        for (int ri = 0; ri < NRI; ++ri) {
            const int I = c.wave + 4 * ri;
            if (I <= CH) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int pl = 4 * c.q + reg;
                    pb[(16 * I + pl) * PLD + c.r16] = u_entry<NT>(T[ri][CH][reg], I, CH, pl, c);
                }
            }
        }
        __syncthreads();
        if (c.tid < 16 * (CH + 1)) {
            const f32x4* urow = reinterpret_cast<const f32x4*>(&pb[c.tid * PLD]);
            const f32x4* wch = reinterpret_cast<const f32x4*>(&wv[16 * CH]);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 u = urow[v], w = wch[v];
                alpha_p += u[0] * w[0] + u[1] * w[1] + u[2] * w[2] + u[3] * w[3];
            }
        }
        if constexpr (WANT_GRAD) {
            const float* base = pb + c.r16 * PLD + 4 * c.q;
            const int ra = w_row_a<NT>(c.wave), rb = w_row_b<NT>(c.wave);
            if (ra >= 0 && ra <= CH) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(base + ra * 16 * PLD) * ncoef;
#pragma unroll
                for (int tj = 0; tj < NT; ++tj) {
                    if (tj <= ra) {
                        const f32x4 bf = *reinterpret_cast<const f32x4*>(base + tj * 16 * PLD);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bf[t], acc[tj], 0, 0, 0);
                    }
                }
            }
            if constexpr (NT > 4) {
                if (rb >= 0 && rb <= CH) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(base + rb * 16 * PLD) * ncoef;
#pragma unroll
                    for (int tj = 0; tj < NT - 4; ++tj) {
                        if (tj <= rb) {
                            const f32x4 bf = *reinterpret_cast<const f32x4*>(base + tj * 16 * PLD);
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                acc[NT + tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bf[t], acc[NT + tj], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
        blk_u_pass<NT, CH + 1, WANT_GRAD>(T, acc, pb, wv, alpha_p, ncoef, c);
    }
}

__device__ __forceinline__ void blk_sum5(float (&v)[5], float* red) {
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 5; ++i) red[(threadIdx.x >> 6) * 5 + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; ++i) v[i] = red[i] + red[5 + i] + red[10 + i] + red[15 + i];
}

// W tile row `row` (scalar), slots acc[0 .. nslots): rank-1 term and final store
template <int NT>
__device__ __forceinline__ void w_rank1_row(f32x4* acc, int row, int nslots, const float* al, float coef, const BlkCtx<NT>& c) {
    const f32x4 ar = *reinterpret_cast<const f32x4*>(&al[16 * row + 4 * c.q]) * coef;
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        if (tj < nslots && tj <= row) {
            const float ac = al[16 * tj + c.r16];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[tj][reg] = __builtin_fmaf(ar[reg], ac, acc[tj][reg]);
        }
    }
}

template <int NT>
__device__ __forceinline__ void w_store_row(const f32x4* acc, int row, int nslots, brsrc Wr, const BlkCtx<NT>& c) {
#pragma unroll
    for (int tj = 0; tj < NT; ++tj) {
        if (tj < nslots && tj <= row) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int pl = 4 * c.q + reg;
                bool ok = true;
                if (row == NT - 1) ok = ok && (pl < c.tyN);                 // gi < N
                if (tj == NT - 1) ok = ok && c.col_ok;                      // gj < N
                if (tj == row) ok = ok && (c.r16 <= pl);                    // lower triangle of the diagonal tile
                const float v = acc[tj][reg];
                if (ok) {
                    bstore(Wr, v, c.vo_rc, ((16 * row + reg) * c.N + 16 * tj) * 4);
                    if (!(tj == row && c.r16 == pl)) bstore(Wr, v, c.vo_cr, (16 * tj * c.N + 16 * row + reg) * 4);
                }
            }
        }
    }
}

template <int NT, bool WANT_GRAD, bool WANT_CHOL>
#ifndef DKT_BLK_MINW
#define DKT_BLK_MINW 3
#endif
__global__ __launch_bounds__(256, ((WANT_GRAD && WANT_CHOL) ? 2 : DKT_BLK_MINW)) void mll_blk_kernel(MllArgs a) {
    constexpr int NP = 16 * NT;
    constexpr int NACC = acc_slots<NT>();
    __shared__ __attribute__((aligned(16))) float pb[NP * PLD];
    __shared__ __attribute__((aligned(16))) float colbuf[2 * NP];
    __shared__ __attribute__((aligned(16))) float wv[NP];
    __shared__ __attribute__((aligned(16))) float al[NP];
    __shared__ float red[20];

    const int b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63;
    const int N = a.N, C = a.C;
    BlkCtx<NT> ctx;
    ctx.N = N; ctx.tyN = N - 16 * (NT - 1); ctx.r16 = lane & 15; ctx.q = lane >> 4; ctx.tid = tid;
    ctx.wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // provably wave-uniform: scalar branches on the tile-row index
    ctx.col_ok = ctx.r16 < ctx.tyN;
    ctx.vo_rc = (4 * ctx.q * N + ctx.r16) * 4;
    ctx.vo_cr = (ctx.r16 * N + 4 * ctx.q) * 4;
    const brsrc Er = make_rsrc(a.E + (size_t)b * N * N, N * N * 4);
    const int ra = w_row_a<NT>(ctx.wave), rb = w_row_b<NT>(ctx.wave);

    f32x4 T[NRI][NT];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool poisoned = false;

    for (int c = 0; c < C; ++c) {
        const float svc = a.sv[c], mc = a.mean[c], nzc = a.noise[c];
        const brsrc yr = make_rsrc(a.Y + (size_t)b * a.y_bstride + (size_t)c * N, N * 4);
        float log2sum = 0.f;
        int fail_at = 0;
        float jit = 0.f;
        for (int attempt = 0; attempt <= a.max_tries; ++attempt) {
            jit = 0.f;
            if (attempt > 0) {
                jit = a.jitter0;
                for (int i = 1; i < attempt; ++i) jit *= 10.f;
            }
            blk_form<NT>(T, Er, yr, svc, mc, nzc + jit, ctx);
            __syncthreads();          // previous users of colbuf / pb / wv / al are done
            log2sum = 0.f;
            fail_at = blk_all<NT, 0>(T, colbuf, pb, log2sum, ctx);
            if (fail_at == 0) break;
        }
        const size_t bc = (size_t)b * C + c;
        if (fail_at != 0) {
            const float qnan = __int_as_float(0x7fc00000);
            if (tid == 0) {
                a.logp[bc] = qnan;
                a.jitter_used[bc] = jit;
                a.info[bc] = fail_at;
                if (WANT_GRAD) { a.dsv[bc] = qnan; a.dmean[bc] = qnan; a.dnoise[bc] = qnan; }
            }
            for (int i = tid; i < N; i += 256) a.alpha[bc * N + i] = qnan;
            if (WANT_CHOL)
                for (int idx = tid; idx < N * N; idx += 256) a.L[bc * N * N + idx] = qnan;
            poisoned = true;
            __syncthreads();
            continue;
        }
        // ---- w row -> LDS, |w|^2, |U|_F^2 ----
        float v5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};     // quad, |U|_F^2, sum alpha, sum alpha^2, -
        if (tid < NP) wv[tid] = 0.f;
        __syncthreads();
        {
            const int qn = ctx.tyN >> 2, rn = ctx.tyN & 3;
#pragma unroll
            for (int ri = 0; ri < NRI; ++ri) {
                const int I = ctx.wave + 4 * ri;
                if (I < NT) {
#pragma unroll
                    for (int J = 0; J < NT; ++J) {
                        const f32x4 t = T[ri][J];
                        if (I == NT - 1) {                                   // the w row lives in the last block row
                            float x = (rn == 0) ? t[0] : (rn == 1) ? t[1] : (rn == 2) ? t[2] : t[3];
                            if (J == NT - 1) x = ctx.col_ok ? x : 0.f;
                            if (ctx.q == qn) {
                                wv[16 * J + ctx.r16] = x;
                                v5[0] += x * x;
                            }
                        }
                        if (J >= I) {
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) {
                                const float u = u_entry<NT>(t[reg], I, J, 4 * ctx.q + reg, ctx);
                                v5[1] += u * u;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        // ---- alpha = U w (and, training, the U U^T part of W) chunk by chunk ----
        const float cw = (WANT_GRAD && a.cls_weight) ? a.cls_weight[c] : 1.0f;
        const float coef = 0.5f * cw * svc;
        float alpha_p = 0.f;
        blk_u_pass<NT, 0, WANT_GRAD>(T, acc, pb, wv, alpha_p, -coef, ctx);
        if (tid >= N) alpha_p = 0.f;
        if (tid < NP) al[tid] = alpha_p;
        if (tid < N) a.alpha[bc * N + tid] = alpha_p;
        v5[2] = alpha_p;
        v5[3] = alpha_p * alpha_p;
        blk_sum5(v5, red);                             // (its barriers also publish al[])
        const float quad = v5[0], trk = v5[1], asum = v5[2], a2 = v5[3];
        if (tid == 0) {
            a.logp[bc] = -0.5f * quad - 0.34657359027997264f * log2sum - (float)N * DKT_HALF_LOG_2PI;
            a.jitter_used[bc] = jit;
            a.info[bc] = 0;
            if (WANT_GRAD) {
                const float nz_eff = nzc + jit;
                a.dmean[bc] = asum;
                a.dnoise[bc] = 0.5f * (a2 - trk);
                a.dsv[bc] = 0.5f * ((quad - (float)N) - nz_eff * (a2 - trk)) / svc;
            }
        }
        if constexpr (WANT_CHOL) {
            const brsrc Lr = make_rsrc(a.L + bc * N * N, N * N * 4);
#pragma unroll
            for (int ri = 0; ri < NRI; ++ri) {
                const int I = ctx.wave + 4 * ri;
                if (I < NT) {
#pragma unroll
                    for (int J = 0; J < NT; ++J) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int pl = 4 * ctx.q + reg;
                            bool ok = true;
                            if (I == NT - 1) ok = ok && (pl < ctx.tyN);
                            if (J == NT - 1) ok = ok && ctx.col_ok;
                            float x = 0.f;
                            if (I > J) x = T[ri][J][reg];
                            else if (I == J) x = (pl > ctx.r16) ? T[ri][J][reg] : ((pl == ctx.r16) ? 1.0f / T[ri][J][reg] : 0.f);
                            if (ok) bstore(Lr, x, ctx.vo_rc, ((16 * I + reg) * N + 16 * J) * 4);
                        }
                    }
                }
            }
        }
        if constexpr (WANT_GRAD) {
            if (ra >= 0) w_rank1_row<NT>(acc, ra, NT, al, coef, ctx);
            if constexpr (NT > 4) {
                if (rb >= 0) w_rank1_row<NT>(acc + NT, rb, NT - 4, al, coef, ctx);
            }
        }
        __syncthreads();
    }

    if constexpr (WANT_GRAD) {
        float* Wb = a.W + (size_t)b * N * N;
        if (poisoned) {
            const float qnan = __int_as_float(0x7fc00000);
            for (int idx = tid; idx < N * N; idx += 256) Wb[idx] = qnan;
        } else {
            const brsrc Wr = make_rsrc(Wb, N * N * 4);
            if (ra >= 0) w_store_row<NT>(acc, ra, NT, Wr, ctx);
            if constexpr (NT > 4) {
                if (rb >= 0) w_store_row<NT>(acc + NT, rb, NT - 4, Wr, ctx);
            }
        }
    }
}

template <int NT>
void launch_blk(const MllArgs& a, hipStream_t st) {
    const bool g = (a.flags & DKT_MLL_WANT_GRAD) != 0, c = (a.flags & DKT_MLL_WANT_CHOL) != 0;
    if (g && c) hipLaunchKernelGGL((mll_blk_kernel<NT, true, true>), dim3(a.B), dim3(256), 0, st, a);
    else if (g) hipLaunchKernelGGL((mll_blk_kernel<NT, true, false>), dim3(a.B), dim3(256), 0, st, a);
    else if (c) hipLaunchKernelGGL((mll_blk_kernel<NT, false, true>), dim3(a.B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((mll_blk_kernel<NT, false, false>), dim3(a.B), dim3(256), 0, st, a);
}

}  // namespace

bool dkt_mll_blk_launch(const MllArgs& a, hipStream_t st) {
    const char* env = getenv("DKT_MLL_BLK");           // off by default: parity-green but slower than the register kernel
    if (!env || atoi(env) == 0) return false;           // (same number of barrier-separated steps)
    const int nt = (a.N + 1 + 15) / 16;
    switch (nt) {
        case 2: launch_blk<2>(a, st); return true;
        case 3: launch_blk<3>(a, st); return true;
        case 4: launch_blk<4>(a, st); return true;
        case 5: launch_blk<5>(a, st); return true;
        case 6: launch_blk<6>(a, st); return true;
        case 7: launch_blk<7>(a, st); return true;
        case 8: launch_blk<8>(a, st); return true;
        default: return false;
    }
}
