// dkt_mll_wave.hip -- WAVE-PER-EPISODE exact-GP marginal likelihood for N + 1 <= 112: no barriers at all.
//
// One 64-lane wave owns one episode and runs its C class models in sequence.  The (N+1) x N working
// matrix (L below the diagonal, U = L^-T above, w = L^-1 r in row N -- see dkt_mll.hip) is distributed
// 2-D cyclically over an 8 x 8 lane grid,
//       lane (ty, tx) owns  Mw[ty + 8*pi][tx + 8*ji],  pi, ji in [0, NB),  NB = ceil((N+1)/8)   (196 VGPRs at N = 105).
// A sweep step needs no LDS memory and no s_barrier: the pivot comes through v_readlane, the row / column
// factors through ds_bpermute_b32 (LDS crossbar only), everything else is lane-local v_fma.  Compared with
// the 4-wave register kernel (dkt_mll_reg.hip) the per-step bookkeeping is paid once per matrix instead of
// once per wave: ~150 instead of ~520 wave-instructions per sweep step.
//
// Gradient: W = sum_c coef_c (alpha alpha^T - U U^T) via v_mfma_f32_16x16x4_f32 in 4 passes of <= 7 tiles
// (28 accumulator VGPRs) per class, the wave's private LDS chunk buffer re-filled per pass, accumulated
// into W[b] in memory (the same lanes touch the same words class after class: no atomics, deterministic).
#include "dkt_mll.h"

namespace {

constexpr int WULD = 24;          // LDS row stride of a 16-column chunk (floats)

__device__ __forceinline__ float bperm(int byte_addr, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int NB>
struct WCtx {
    int N, tx, ty, lane;
    bool col_ok, lower_eq;
};

template <int NB, int KQ>
__device__ __forceinline__ int wsweep_block(float (&A)[NB][NB], float& log2sum, const WCtx<NB>& c) {
    const int kend = min(8, c.N - 8 * KQ);
    const int tx = c.tx, ty = c.ty;
    for (int kr = 0; kr < kend; ++kr) {
        const float d = readlane_f(A[KQ][KQ], kr * 9);              // pivot lives at lane (kr, kr)
        if (!(d > 0.f)) return 8 * KQ + kr + 1;
        const float rinv = __builtin_amdgcn_rsqf(d);
        log2sum += __builtin_amdgcn_logf(d);
        const int src_row = ((c.lane & 56) | kr) << 2;              // lane (ty, kr)
        const int src_col = ((tx << 3) | kr) << 2;                  // lane (tx, kr): row index = my column index
        float cp[NB];
#pragma unroll
        for (int pi = 0; pi < NB; ++pi) cp[pi] = bperm(src_row, A[pi][KQ]) * rinv;
        cp[KQ] = (ty == kr) ? rinv : cp[KQ];
        const bool row_le_k = ty <= kr;
        const float cpk_le = row_le_k ? cp[KQ] : 0.f;
        const float cpk_dd = (row_le_k || c.lower_eq) ? cp[KQ] : 0.f;
#pragma unroll
        for (int ji = KQ; ji < NB; ++ji) {
            float cj = bperm(src_col, A[ji][KQ]) * rinv;
            if (ji == KQ) cj = (tx > kr) ? cj : 0.f;
            if (ji == NB - 1) cj = c.col_ok ? cj : 0.f;
#pragma unroll
            for (int pi = 0; pi < NB; ++pi) {
                if (pi < KQ) {
                    A[pi][ji] = __builtin_fmaf(-cp[pi], cj, A[pi][ji]);
                } else if (pi == KQ) {
                    A[pi][ji] = __builtin_fmaf(-(ji == KQ ? cpk_dd : cpk_le), cj, A[pi][ji]);
                } else if (pi > ji) {
                    A[pi][ji] = __builtin_fmaf(-cp[pi], cj, A[pi][ji]);
                } else if (pi == ji) {
                    A[pi][ji] = __builtin_fmaf(-(c.lower_eq ? cp[pi] : 0.f), cj, A[pi][ji]);
                }
            }
        }
        const bool own = tx == kr;                                   // finalise column k in place
#pragma unroll
        for (int pi = 0; pi < NB; ++pi) A[pi][KQ] = own ? cp[pi] : A[pi][KQ];
    }
    return 0;
}

template <int NB, int KQ>
__device__ __forceinline__ int wsweep_all(float (&A)[NB][NB], float& log2sum, const WCtx<NB>& c) {
    if constexpr (KQ < NB) {
        if (8 * KQ >= c.N) return 0;
        const int f = wsweep_block<NB, KQ>(A, log2sum, c);
        if (f) return f;
        return wsweep_all<NB, KQ + 1>(A, log2sum, c);
    } else {
        return 0;
    }
}

// one pass of the W product: tile rows [R0, R1] (16-row MFMA tiles), all 16-column chunks.
template <int NB, int NT16, int R0, int R1, int CH>
__device__ __forceinline__ void wprod_pass(f32x4* acc, const float (&A)[NB][NB], const float (&alpha)[NB], float* ub,
                                           const WCtx<NB>& c, bool row_ok, bool is_acol, float coef) {
    if constexpr (CH < NT16) {
        // chunk CH = columns [16 CH, 16 CH + 16) = 8-blocks ji = 2 CH, 2 CH + 1; rows p = ty + 8 pi
        if constexpr (CH >= R0) {                                    // row blocks > CH are zero in this chunk: skip early chunks
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                constexpr int dummy = 0; (void)dummy;
                const int ji = 2 * CH + h;
#pragma unroll
                for (int pi = 0; pi < NB; ++pi) {
                    float v = 0.f;
                    if (ji < NB) {
                        if (ji > pi) v = A[pi][ji];
                        else if (ji == pi) v = (c.tx >= c.ty) ? A[pi][ji] : 0.f;
                        if (ji == NB - 1) {
                            v = c.col_ok ? v : 0.f;
                            v = is_acol ? alpha[pi] : v;
                        }
                        if (pi == NB - 1) v = row_ok ? v : 0.f;
                    }
                    ub[(c.ty + 8 * pi) * WULD + c.tx + 8 * h] = v;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): the wave's own LDS writes have landed
            __builtin_amdgcn_sched_barrier(0);
            const int r16 = c.lane & 15, q = c.lane >> 4;
            const float* base = ub + r16 * WULD + 4 * q;
            f32x4 sc;
#pragma unroll
            for (int t = 0; t < 4; ++t) sc[t] = (16 * CH + 4 * q + t == c.N) ? coef : -coef;
            int ai = 0;
#pragma unroll
            for (int ti = R0; ti <= R1; ++ti) {
                if (ti <= CH) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(base + ti * 16 * WULD) * sc;
#pragma unroll
                    for (int tj = 0; tj <= ti; ++tj) {
                        const f32x4 bf = *reinterpret_cast<const f32x4*>(base + tj * 16 * WULD);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            acc[ai + tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bf[t], acc[ai + tj], 0, 0, 0);
                    }
                }
                ai += ti + 1;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                      // fragment reads done before the buffer is refilled
        }
        wprod_pass<NB, NT16, R0, R1, CH + 1>(acc, A, alpha, ub, c, row_ok, is_acol, coef);
    }
}

template <int R0, int R1>
__device__ __forceinline__ void wprod_store(const f32x4* acc, float* Wb, int N, int lane, bool first) {
    const int r16 = lane & 15, q = lane >> 4;
    int ai = 0;
#pragma unroll
    for (int ti = R0; ti <= R1; ++ti) {
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gi = ti * 16 + 4 * q + reg, gj = tj * 16 + r16;
                if (gi < N && gj < N && gj <= gi) {
                    float v = acc[ai + tj][reg];
                    if (!first) v += Wb[gi * N + gj];
                    Wb[gi * N + gj] = v;
                    if (gi != gj) Wb[gj * N + gi] = v;
                }
            }
        }
        ai += ti + 1;
    }
}

template <int NB, int NT16, int R0, int R1>
__device__ __forceinline__ void wprod_rows(const float (&A)[NB][NB], const float (&alpha)[NB], float* ub, float* Wb,
                                           const WCtx<NB>& c, bool row_ok, bool is_acol, float coef, bool first) {
    constexpr int NACC = (R1 + 1) * (R1 + 2) / 2 - R0 * (R0 + 1) / 2;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    wprod_pass<NB, NT16, R0, R1, 0>(acc, A, alpha, ub, c, row_ok, is_acol, coef);
    wprod_store<R0, R1>(acc, Wb, c.N, c.lane, first);
}

template <int NB, bool WANT_GRAD, bool WANT_CHOL>
__global__ __launch_bounds__(64, 1) void mll_wave_kernel(MllArgs a) {
    constexpr int NT16 = (NB + 1) / 2;
    __shared__ __attribute__((aligned(16))) float ubuf[WANT_GRAD ? 16 * NT16 * WULD : 4];

    const int b = blockIdx.x, lane = threadIdx.x;
    const int tx = lane & 7, ty = lane >> 3;
    const int N = a.N, C = a.C;
    const int tyN = N - 8 * (NB - 1);
    const bool lower_eq = ty >= tx, upper_eq = tx >= ty;
    const bool row_ok = ty < tyN, is_w = ty == tyN;
    const bool col_ok = tx < tyN, is_acol = tx == tyN;
    const float* Eb = a.E + (size_t)b * N * N;
    WCtx<NB> ctx;
    ctx.N = N; ctx.tx = tx; ctx.ty = ty; ctx.lane = lane; ctx.col_ok = col_ok; ctx.lower_eq = lower_eq;
    bool poisoned = false;

    for (int c = 0; c < C; ++c) {
        const float svc = a.sv[c], mc = a.mean[c], nzc = a.noise[c];
        const float* yc = a.Y + (size_t)b * a.y_bstride + (size_t)c * N;
        float A[NB][NB];
        float log2sum = 0.f;
        int fail_at = 0;
        float jit = 0.f;
        for (int attempt = 0; attempt <= a.max_tries; ++attempt) {
            jit = 0.f;
            if (attempt > 0) {
                jit = a.jitter0;
                for (int i = 1; i < attempt; ++i) jit *= 10.f;
            }
#pragma unroll
            for (int pi = 0; pi < NB; ++pi) {
#pragma unroll
                for (int ji = 0; ji < NB; ++ji) {
                    const int p = ty + 8 * pi, j = tx + 8 * ji;
                    float v = 0.f;
                    if (pi >= ji) {
                        bool ld = true;
                        if (pi == ji) ld = lower_eq;
                        if (pi == NB - 1) ld = ld && row_ok;
                        if (ji == NB - 1) ld = ld && col_ok;
                        if (ld) {
                            v = svc * Eb[p * N + j];
                            if (pi == ji && tx == ty) v += nzc + jit;
                        }
                        if (pi == NB - 1) {
                            bool lw = is_w;
                            if (ji == NB - 1) lw = lw && col_ok;
                            if (lw) v = yc[j] - mc;
                        }
                    }
                    A[pi][ji] = v;
                }
            }
            log2sum = 0.f;
            fail_at = wsweep_all<NB, 0>(A, log2sum, ctx);
            if (fail_at == 0) break;
        }
        const size_t bc = (size_t)b * C + c;
        if (fail_at != 0) {
            const float qnan = __int_as_float(0x7fc00000);
            if (lane == 0) {
                a.logp[bc] = qnan;
                a.jitter_used[bc] = jit;
                a.info[bc] = fail_at;
                if (WANT_GRAD) { a.dsv[bc] = qnan; a.dmean[bc] = qnan; a.dnoise[bc] = qnan; }
            }
            for (int i = lane; i < N; i += 64) a.alpha[bc * N + i] = qnan;
            if (WANT_CHOL)
                for (int idx = lane; idx < N * N; idx += 64) a.L[bc * N * N + idx] = qnan;
            poisoned = true;
            continue;
        }
        // ---- w (row N, held by lanes ty == tyN) to every lane of the same column class tx ----
        float wj[NB];
        const int src_w = ((tyN << 3) | tx) << 2;
#pragma unroll
        for (int ji = 0; ji < NB; ++ji) wj[ji] = bperm(src_w, A[NB - 1][ji]);
        if (!col_ok) wj[NB - 1] = 0.f;
        float quad = 0.f;
#pragma unroll
        for (int ji = 0; ji < NB; ++ji) quad += wj[ji] * wj[ji];          // per tx: partial over its columns
        float alpha[NB];
        float asum = 0.f, a2 = 0.f, trk = 0.f;
#pragma unroll
        for (int pi = 0; pi < NB; ++pi) {
            float s = 0.f, u2 = 0.f;
#pragma unroll
            for (int ji = pi; ji < NB; ++ji) {
                float u = A[pi][ji];
                if (ji == pi) u = upper_eq ? u : 0.f;
                if (ji == NB - 1) u = col_ok ? u : 0.f;
                s += u * wj[ji];
                u2 += u * u;
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) s += __shfl_xor(s, o, DKT_WAVE);
            if (pi == NB - 1) { s = row_ok ? s : 0.f; u2 = row_ok ? u2 : 0.f; }
            alpha[pi] = s;
            trk += u2;
            if (tx == 0) {
                asum += s;
                a2 += s * s;
                if (pi < NB - 1 || row_ok) a.alpha[bc * N + ty + 8 * pi] = s;
            }
        }
        // wave-wide sums: quad over tx (rows identical across ty: take ty == 0), the rest over all lanes
        quad = (ty == 0) ? quad : 0.f;
        quad = wave_allsum(quad);
        asum = wave_allsum(asum);
        a2 = wave_allsum(a2);
        trk = wave_allsum(trk);
        if (lane == 0) {
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
        if (WANT_CHOL) {
            float* Lb = a.L + bc * N * N;
#pragma unroll
            for (int pi = 0; pi < NB; ++pi) {
#pragma unroll
                for (int ji = 0; ji < NB; ++ji) {
                    const int p = ty + 8 * pi, j = tx + 8 * ji;
                    bool ok = true;
                    if (pi == NB - 1) ok = ok && row_ok;
                    if (ji == NB - 1) ok = ok && col_ok;
                    if (ok) {
                        float v = 0.f;
                        if (pi > ji) v = A[pi][ji];
                        else if (pi == ji) v = (ty > tx) ? A[pi][ji] : ((ty == tx) ? 1.0f / A[pi][ji] : 0.f);
                        Lb[p * N + j] = v;
                    }
                }
            }
        }
        if (WANT_GRAD) {
            const float cw = a.cls_weight ? a.cls_weight[c] : 1.0f;
            const float coef = 0.5f * cw * svc;
            float* Wb = a.W + (size_t)b * N * N;
            const bool first = (c == 0);
            // 4 passes of tile rows; <= 7 accumulator tiles each
            if constexpr (NT16 >= 7) {
                wprod_rows<NB, NT16, 0, 2>(A, alpha, ubuf, Wb, ctx, row_ok, is_acol, coef, first);
                wprod_rows<NB, NT16, 3, 3>(A, alpha, ubuf, Wb, ctx, row_ok, is_acol, coef, first);
                wprod_rows<NB, NT16, 4, 4>(A, alpha, ubuf, Wb, ctx, row_ok, is_acol, coef, first);
                wprod_rows<NB, NT16, 5, 5>(A, alpha, ubuf, Wb, ctx, row_ok, is_acol, coef, first);
                wprod_rows<NB, NT16, 6, NT16 - 1>(A, alpha, ubuf, Wb, ctx, row_ok, is_acol, coef, first);
            } else if constexpr (NT16 >= 4) {
                wprod_rows<NB, NT16, 0, 2>(A, alpha, ubuf, Wb, ctx, row_ok, is_acol, coef, first);
                wprod_rows<NB, NT16, 3, NT16 - 1>(A, alpha, ubuf, Wb, ctx, row_ok, is_acol, coef, first);
            } else {
                wprod_rows<NB, NT16, 0, NT16 - 1>(A, alpha, ubuf, Wb, ctx, row_ok, is_acol, coef, first);
            }
        }
    }
    if (WANT_GRAD && poisoned) {
        float* Wb = a.W + (size_t)b * N * N;
        const float qnan = __int_as_float(0x7fc00000);
        for (int idx = lane; idx < N * N; idx += 64) Wb[idx] = qnan;
    }
}

template <int NB>
void launch_wave(const MllArgs& a, hipStream_t st) {
    const bool g = (a.flags & DKT_MLL_WANT_GRAD) != 0, c = (a.flags & DKT_MLL_WANT_CHOL) != 0;
    if (g && c) hipLaunchKernelGGL((mll_wave_kernel<NB, true, true>), dim3(a.B), dim3(64), 0, st, a);
    else if (g) hipLaunchKernelGGL((mll_wave_kernel<NB, true, false>), dim3(a.B), dim3(64), 0, st, a);
    else if (c) hipLaunchKernelGGL((mll_wave_kernel<NB, false, true>), dim3(a.B), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((mll_wave_kernel<NB, false, false>), dim3(a.B), dim3(64), 0, st, a);
}

}  // namespace

bool dkt_mll_wave_launch(const MllArgs& a, hipStream_t st) {
    const char* env = getenv("DKT_MLL_WAVE");          // off by default: 1 wave/SIMD at N = 105 is slower than the
    if (!env || atoi(env) == 0) return false;          // 4-wave register kernel; kept parity-tested
    const int nb = (a.N + 1 + 7) / 8;
    switch (nb) {
        case 14: launch_wave<14>(a, st); return true;
        default: return false;
    }
}
