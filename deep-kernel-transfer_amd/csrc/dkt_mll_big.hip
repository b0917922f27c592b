// dkt_mll_big.hip -- exact-GP marginal likelihood for N > 127 (the 20-way shapes: N = 320 / 420, "batched Cholesky
// stress" of BASELINE.json), built from blocks the small-N machinery already masters:
//   * every (episode, class) matrix K = sv E + (noise + jitter) I is processed in nbk x nbk blocks of nb <= 111 rows,
//   * diagonal blocks are factored AND inverted by the register-resident sweep (chol_inv_block_kernel in dkt_mll_reg.hip:
//     L_jj below, U_jj = L_jj^-T above), so the panel step needs no triangular solve:  L_ij = A_ij U_jj  is a GEMM,
//   * trailing updates, V = L^-T (block back-substitution with the explicit U_jj) and K^-1 = V V^T are batched fp32-MFMA
//     GEMMs over all B x C matrices of a chunk (one launch per block operation, independent of C),
//   * alpha = V (V^T r), log det from the diagonal of L, the scalar identities of dkt_mll_reg.hip for the hyper-gradients,
//   * W[b] = sum_c coef_c (alpha_c alpha_c^T - K_c^-1) summed over the classes in a fixed order (deterministic).
// The blocked pass runs WITHOUT jitter (attempt 0 of GPyTorch's psd_safe_cholesky); an episode with a matrix that fails it is redone
// -- full retry ladder, per matrix -- by the generic kernel in a fix-up launch that exits at once for every other episode:
// no host read-back, the call stays asynchronous on its stream.
// Replaces the same reference lines as dkt_mll.hip (methods/DKT.py:161-163,177,187,252-254,265,330).
#include "dkt_mll.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Batched strided GEMM on sub-blocks:  C = beta C + alpha op(A) op(B),  op(A): M x K, op(B): K x N.
struct GemmOp {
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    long sA, sB, sC;            // batch strides (floats)
    float alpha, beta;
    bool vecA, vecB;            // 16-byte loads legal for A / B
};

constexpr int GLD = 24;         // LDS row stride (floats) of a [64][16] tile: 6 sixteen-byte units -> conflict-free b128 fragment reads

// One 64 x 16 operand tile: global -> register (tile_load) and register -> LDS as T[row][k] (tile_store), so that the loads of
// the NEXT K slice are in flight while the current one is multiplied.  `TR` = the operand's memory has the tile ROW index as the
// fast index (op = transpose): the thread then loads 4 consecutive rows of one k and scatters them; otherwise 4 consecutive k
// of one row.  vec: 16-byte loads are legal (leading dimension, base and offsets multiples of 4 floats).
template <bool TR>
__device__ __forceinline__ float4 tile_load(const float* __restrict__ P, int ld, int r0, int nrows, int k0, int K, int tid, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!TR) {
        const int r = tid >> 2, k = 4 * (tid & 3);
        if (r0 + r < nrows) {
            const float* p = P + (size_t)(r0 + r) * ld + k0 + k;
            if (vec && k0 + k + 3 < K) v = *reinterpret_cast<const float4*>(p);
            else {
                if (k0 + k + 0 < K) v.x = p[0];
                if (k0 + k + 1 < K) v.y = p[1];
                if (k0 + k + 2 < K) v.z = p[2];
                if (k0 + k + 3 < K) v.w = p[3];
            }
        }
    } else {
        const int k = tid >> 4, r = 4 * (tid & 15);
        if (k0 + k < K) {
            const float* p = P + (size_t)(k0 + k) * ld + r0 + r;
            if (vec && r0 + r + 3 < nrows) v = *reinterpret_cast<const float4*>(p);
            else {
                if (r0 + r + 0 < nrows) v.x = p[0];
                if (r0 + r + 1 < nrows) v.y = p[1];
                if (r0 + r + 2 < nrows) v.z = p[2];
                if (r0 + r + 3 < nrows) v.w = p[3];
            }
        }
    }
    return v;
}

template <bool TR>
__device__ __forceinline__ void tile_store(float* T, const float4& v, int tid) {
    if (!TR) {
        const int r = tid >> 2, k = 4 * (tid & 3);
        *reinterpret_cast<float4*>(T + r * GLD + k) = v;
    } else {
        const int k = tid >> 4, r = 4 * (tid & 15);
        T[(r + 0) * GLD + k] = v.x;
        T[(r + 1) * GLD + k] = v.y;
        T[(r + 2) * GLD + k] = v.z;
        T[(r + 3) * GLD + k] = v.w;
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void bgemm_kernel(GemmOp g) {
    __shared__ __attribute__((aligned(16))) float As[2][64 * GLD], Bs[2][64 * GLD];
    const int bz = blockIdx.z;
    const float* A = g.A + (size_t)bz * g.sA;
    const float* B = g.B + (size_t)bz * g.sB;
    float* C = g.C + (size_t)bz * g.sC;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // op(A)[m][k] is A[m][k] (rows = m, fast index k) or A[k][m] (TA: fast index m);  op(B)^T[n][k] is B[n][k] (TB) or B[k][n]
    float4 ra = tile_load<TA>(A, g.lda, m0, g.M, 0, g.K, tid, g.vecA);
    float4 rb = tile_load<!TB>(B, g.ldb, n0, g.N, 0, g.K, tid, g.vecB);
    tile_store<TA>(As[0], ra, tid);
    tile_store<!TB>(Bs[0], rb, tid);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < g.K; k0 += 16, buf ^= 1) {
        const bool more = k0 + 16 < g.K;
        if (more) {                                     // next slice: loads fly during this slice's MFMAs
            ra = tile_load<TA>(A, g.lda, m0, g.M, k0 + 16, g.K, tid, g.vecA);
            rb = tile_load<!TB>(B, g.ldb, n0, g.N, k0 + 16, g.K, tid, g.vecB);
        }
        // lane group q holds k = 4q .. 4q+3 of the 16-wide slice (one b128 read); the t-th MFMA contracts k = {t, 4+t, 8+t, 12+t}
        const f32x4 a = *reinterpret_cast<const f32x4*>(As[buf] + (16 * wave + r16) * GLD + 4 * q);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const f32x4 bfr = *reinterpret_cast<const f32x4*>(Bs[buf] + (16 * ct + r16) * GLD + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], bfr[t], acc[ct], 0, 0, 0);
        }
        if (more) {                                     // the other buffer was last read before the previous barrier
            tile_store<TA>(As[buf ^ 1], ra, tid);
            tile_store<!TB>(Bs[buf ^ 1], rb, tid);
        }
        __syncthreads();
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int m = m0 + 16 * wave + 4 * q + reg, n = n0 + 16 * ct + r16;
            if (m < g.M && n < g.N) {
                float* c = C + (size_t)m * g.ldc + n;
                const float v = g.alpha * acc[ct][reg];
                *c = (g.beta == 0.f) ? v : __builtin_fmaf(g.beta, *c, v);
            }
        }
    }
}

inline bool vec_ok(const float* p, int ld, long stride) { return ((uintptr_t)p & 15) == 0 && (ld & 3) == 0 && (stride & 3) == 0; }

void gemm(hipStream_t st, int nmat, bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda, long sA,
          const float* B, int ldb, long sB, float beta, float* C, int ldc, long sC) {
    GemmOp g{A, B, C, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha, beta, vec_ok(A, lda, sA), vec_ok(B, ldb, sB)};
    dim3 grid((N + 63) / 64, (M + 63) / 64, nmat), blk(256);
    if (!ta && !tb) hipLaunchKernelGGL((bgemm_kernel<false, false>), grid, blk, 0, st, g);
    else if (!ta && tb) hipLaunchKernelGGL((bgemm_kernel<false, true>), grid, blk, 0, st, g);
    else if (ta && !tb) hipLaunchKernelGGL((bgemm_kernel<true, false>), grid, blk, 0, st, g);
    else hipLaunchKernelGGL((bgemm_kernel<true, true>), grid, blk, 0, st, g);
}

// ---------------------------------------------------------------------------------------------
// Kw[m] = sv_c E[b] + (noise_c + jit[m]) I on the lower block triangle (all the factorisation reads)
__global__ __launch_bounds__(256) void big_form_kernel(const float* __restrict__ E, const float* __restrict__ sv,
                                                       const float* __restrict__ noise, const float* __restrict__ jit,
                                                       float* __restrict__ Kw, int b0, int C, int N, int nb) {
    const int m = blockIdx.y, b = b0 + m / C, c = m % C;
    const float* Eb = E + (size_t)b * N * N;
    float* K = Kw + (size_t)m * N * N;
    const float s = sv[c], dg = noise[c] + jit[m];
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < N * N; idx += gridDim.x * 256) {
        const int i = idx / N, j = idx - i * N;
        if (j / nb <= i / nb) K[idx] = s * Eb[idx] + (i == j ? dg : 0.f);
    }
}

// w = V^T r (TR = true) or alpha = V w (TR = false) with V upper (block-)triangular, one workgroup per matrix.
// TR: a thread owns output i and walks down column i (consecutive threads read consecutive addresses of a row of V);
// non-TR: a wave owns output i and reduces over the row.
template <bool TR>
__global__ __launch_bounds__(256) void big_matvec_kernel(const float* __restrict__ Vm, const float* __restrict__ x, float* __restrict__ y,
                                                         const float* __restrict__ Y, long y_bstride, const float* __restrict__ mean,
                                                         int b0, int C, int N, bool x_is_targets) {
    extern __shared__ float xs[];
    const int m = blockIdx.x, b = b0 + m / C, c = m % C;
    const float* V = Vm + (size_t)m * N * N;
    const float* xv = x_is_targets ? (Y + (size_t)b * y_bstride + (size_t)c * N) : (x + (size_t)m * N);
    const float shift = x_is_targets ? mean[c] : 0.f;
    for (int i = threadIdx.x; i < N; i += 256) xs[i] = xv[i] - shift;
    __syncthreads();
    if (TR) {
        for (int i = threadIdx.x; i < N; i += 256) {    // (V^T r)_i = sum_{k<=i} V_ki r_k
            float s = 0.f;
            for (int k = 0; k <= i; ++k) s = __builtin_fmaf(V[(size_t)k * N + i], xs[k], s);
            y[(size_t)m * N + i] = s;
        }
    } else {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int i = wave; i < N; i += 4) {             // (V w)_i = sum_{k>=i} V_ik w_k
            float s = 0.f;
            for (int k = i + lane; k < N; k += 64) s = __builtin_fmaf(V[(size_t)i * N + k], xs[k], s);
            s = wave_sum(s);
            if (lane == 0) y[(size_t)m * N + i] = s;
        }
    }
}

// per matrix: logp, alpha out, hyper-gradient scalars (tr K^-1 = |V|_F^2), jitter_used
__global__ __launch_bounds__(256) void big_finish_kernel(MllArgs a, const float* __restrict__ Lm, const float* __restrict__ Vm,
                                                         const float* __restrict__ wv, const float* __restrict__ al,
                                                         const float* __restrict__ jit, const int32_t* __restrict__ info_m, int b0) {
    __shared__ float red[8];
    const int m = blockIdx.x, N = a.N, C = a.C, b = b0 + m / C, c = m % C, tid = threadIdx.x;
    const size_t bc = (size_t)b * C + c;
    const float* L = Lm + (size_t)m * N * N;
    const float* V = Vm + (size_t)m * N * N;
    const float* yc = a.Y + (size_t)b * a.y_bstride + (size_t)c * N;
    const int bad = info_m[m];
    if (bad != 0) {
        const float qnan = __int_as_float(0x7fc00000);
        if (tid == 0) {
            a.logp[bc] = qnan; a.jitter_used[bc] = jit[m]; a.info[bc] = bad;
            if (a.flags & DKT_MLL_WANT_GRAD) { a.dsv[bc] = qnan; a.dmean[bc] = qnan; a.dnoise[bc] = qnan; }
        }
        for (int i = tid; i < N; i += 256) a.alpha[bc * N + i] = qnan;
        return;
    }
    float ld = 0.f, quad = 0.f, asum = 0.f, a2 = 0.f, ra = 0.f, vf = 0.f;
    for (int i = tid; i < N; i += 256) {
        ld += logf(L[(size_t)i * N + i]);
        const float w = wv[(size_t)m * N + i], av = al[(size_t)m * N + i];
        quad = __builtin_fmaf(w, w, quad);
        asum += av;
        a2 = __builtin_fmaf(av, av, a2);
        ra = __builtin_fmaf(yc[i] - a.mean[c], av, ra);
        a.alpha[bc * N + i] = av;
    }
    for (int idx = tid; idx < N * N; idx += 256) {
        const int i = idx / N, j = idx - i * N;
        if (j >= i) { const float v = V[idx]; vf = __builtin_fmaf(v, v, vf); }
    }
    ld = block_sum_256(ld, red); quad = block_sum_256(quad, red); asum = block_sum_256(asum, red);
    a2 = block_sum_256(a2, red); ra = block_sum_256(ra, red); vf = block_sum_256(vf, red);
    if (tid == 0) {
        a.logp[bc] = -0.5f * quad - ld - (float)N * DKT_HALF_LOG_2PI;
        a.jitter_used[bc] = jit[m];
        a.info[bc] = 0;
        if (a.flags & DKT_MLL_WANT_GRAD) {
            const float nz_eff = a.noise[c] + jit[m];
            a.dmean[bc] = asum;
            a.dnoise[bc] = 0.5f * (a2 - vf);
            a.dsv[bc] = 0.5f * ((ra - (float)N) - nz_eff * (a2 - vf)) / a.sv[c];
        }
    }
}

// W[b] = sum_c coef_c (alpha_c alpha_c^T - Kinv_c), Kinv given by its lower block triangle (mirrored here); NaN if any class failed
__global__ __launch_bounds__(256) void big_w_kernel(MllArgs a, const float* __restrict__ Kinv, const float* __restrict__ al,
                                                    const int32_t* __restrict__ info_m, int b0, int nb) {
    const int bl = blockIdx.y, b = b0 + bl, N = a.N, C = a.C;
    float* Wb = a.W + (size_t)b * N * N;
    bool poisoned = false;
    for (int c = 0; c < C; ++c) poisoned = poisoned || info_m[bl * C + c] != 0;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < N * N; idx += gridDim.x * 256) {
        const int i = idx / N, j = idx - i * N;
        if (j > i) continue;                              // lower triangle computed, mirrored
        float s = 0.f;
        for (int c = 0; c < C; ++c) {
            const size_t m = (size_t)bl * C + c;
            const float coef = 0.5f * (a.cls_weight ? a.cls_weight[c] : 1.0f) * a.sv[c];
            // Kinv block (bi, bj) with bi >= bj was computed; (i, j) with i >= j lies in such a block or in a diagonal block
            const float kinv = (i / nb >= j / nb) ? Kinv[m * N * N + (size_t)i * N + j] : Kinv[m * N * N + (size_t)j * N + i];
            s += coef * (al[m * N + i] * al[m * N + j] - kinv);
        }
        if (poisoned) s = __int_as_float(0x7fc00000);
        Wb[(size_t)i * N + j] = s;
        Wb[(size_t)j * N + i] = s;
    }
}

// L[b,c] <- lower triangle of Lm (upper zero)
__global__ __launch_bounds__(256) void big_chol_out_kernel(MllArgs a, const float* __restrict__ Lm, const int32_t* __restrict__ info_m, int b0) {
    const int m = blockIdx.y, N = a.N, C = a.C, b = b0 + m / C, c = m % C;
    float* Lo = a.L + ((size_t)b * C + c) * N * N;
    const float* L = Lm + (size_t)m * N * N;
    const bool bad = info_m[m] != 0;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < N * N; idx += gridDim.x * 256) {
        const int i = idx / N, j = idx - i * N;
        Lo[idx] = bad ? __int_as_float(0x7fc00000) : (j <= i ? L[idx] : 0.f);
    }
}

// block size: <= 108 rows (the register sweep needs nb + 1 <= 112), a multiple of 4 so that block origins stay 16-byte aligned
// Block size of the blocked factorisation.  DKT_BIG_NB (measurement aid) overrides the default.
inline int big_nb(int N) {
    static const int forced = [] { const char* v = dkt_variant_env("DKT_BIG_NB"); return v ? atoi(v) : 0; }();
    if (forced >= 16 && forced <= 108) return forced & ~3;
    // blocks of 64 rows = the GEMM tile: no padded MFMA work, and with the merged left-looking launches the extra block columns
    // cost little (N = 420: 7 blocks of 64 beat 4 blocks of 105 by 5 %; N = 320: 5 exact blocks)
    (void)N;
    return 64;
}
constexpr int BIG_CHUNK = 128;          // episodes per chunk of the workspace

inline size_t big_ws_floats(int Bc, int C, int N) {
    const size_t nmat = (size_t)Bc * C, nn = (size_t)N * N;
    return nmat * (4 * nn + 2 * (size_t)N + 4) + 16;       // Kw, Lm, Vm, Kinv/T | w, alpha | jit, attempt, info | counter
}

}  // namespace

size_t dkt_mll_big_workspace_bytes(int B, int C, int N) {
    const int bc = B < BIG_CHUNK ? B : BIG_CHUNK;
    return big_ws_floats(bc, C, N) * sizeof(float);
}

// Returns 0 on success, a negative DKT status otherwise.
int dkt_mll_big_launch(const MllArgs& a, void* workspace, size_t ws_bytes, hipStream_t st) {
    const int N = a.N, C = a.C;
    if (ws_bytes < dkt_mll_big_workspace_bytes(a.B, C, N) || !workspace) return DKT_ERR_WORKSPACE;
    const int nb = big_nb(N), nbk = (N + nb - 1) / nb;
    const long nn = (long)N * N;
    const int Bc = a.B < BIG_CHUNK ? a.B : BIG_CHUNK;
    const size_t nmat_max = (size_t)Bc * C;
    float* Kw = (float*)workspace;
    float* Lm = Kw + nmat_max * nn;
    float* Vm = Lm + nmat_max * nn;
    float* Kinv = Vm + nmat_max * nn;                      // also the temp T of the V recursion (Kinv is formed last)
    float* wv = Kinv + nmat_max * nn;
    float* al = wv + nmat_max * N;
    float* jit = al + nmat_max * N;
    int* attempt_of = (int*)(jit + nmat_max);
    int32_t* info_m = (int32_t*)(attempt_of + nmat_max);
    int* nfail = (int*)(info_m + nmat_max);
    const bool want_grad = (a.flags & DKT_MLL_WANT_GRAD) != 0;
    auto off = [&](int i) { return i * nb; };
    auto sz = [&](int i) { return (i == nbk - 1) ? N - i * nb : nb; };
    auto blk = [&](float* base, int i, int j) { return base + (size_t)off(i) * N + off(j); };

    for (int b0 = 0; b0 < a.B; b0 += Bc) {
        const int bcnt = (a.B - b0 < Bc) ? a.B - b0 : Bc;
        const int nmat = bcnt * C;
        if (hipMemsetAsync(jit, 0, nmat_max * sizeof(float) * 3, st) != hipSuccess) return DKT_ERR_LAUNCH;   // jit (stays 0), attempt_of, info_m
        hipLaunchKernelGGL(big_form_kernel, dim3(32, nmat), dim3(256), 0, st, a.E, a.sv, a.noise, jit, Kw, b0, C, N, nb);
        // ---- blocked Cholesky, LEFT-looking, with explicit inverses of the diagonal blocks.  Per block column j three
        //      launches: (a) all row blocks i >= j at once:  A_ij -= sum_{k<j} L_ik L_jk^T  -- the blocks k < j are contiguous
        //      columns, so this is ONE GEMM with M = N - off(j), K = off(j) (one pass over the result instead of j);
        //      (b) the diagonal block: L_jj and U_jj = L_jj^-T by the register sweep;  (c) the panel below it, all row
        //      blocks at once:  L_ij = A_ij U_jj.
        for (int j = 0; j < nbk; ++j) {
            if (j > 0)
                gemm(st, nmat, false, true, N - off(j), sz(j), off(j), -1.f, blk(Lm, j, 0), N, nn, blk(Lm, j, 0), N, nn, 1.f, blk(Kw, j, j), N, nn);
            dkt_chol_inv_block_launch(blk(Kw, j, j), N, nn, blk(Lm, j, j), N, nn, blk(Vm, j, j), N, nn, sz(j), off(j), info_m, nmat, st);
            if (j + 1 < nbk)
                gemm(st, nmat, false, false, N - off(j + 1), sz(j), sz(j), 1.f, blk(Kw, j + 1, j), N, nn, blk(Vm, j, j), N, nn, 0.f, blk(Lm, j + 1, j), N, nn);
        }
        // ---- V = L^-T, upper block triangular: V_jj = U_jj (already in Vm), V_ij = -U_ii sum_{k=i+1..j} L_ki^T V_kj ----
        for (int j = 1; j < nbk; ++j) {
            for (int i = j - 1; i >= 0; --i) {
                // the row blocks k = i+1 .. j are contiguous in memory: ONE GEMM over the long K instead of a launch (and a
                // read-modify-write of the result) per k
                gemm(st, nmat, true, false, sz(i), sz(j), off(j) + sz(j) - off(i + 1), 1.f, blk(Lm, i + 1, i), N, nn, blk(Vm, i + 1, j), N, nn, 0.f,
                     blk(Kinv, i, j), N, nn);
                gemm(st, nmat, false, false, sz(i), sz(j), sz(i), -1.f, blk(Vm, i, i), N, nn, blk(Kinv, i, j), N, nn, 0.f, blk(Vm, i, j), N, nn);
            }
        }
        // ---- w = V^T r, alpha = V w ----
        hipLaunchKernelGGL((big_matvec_kernel<true>), dim3(nmat), dim3(256), N * sizeof(float), st, Vm, (const float*)nullptr, wv, a.Y, a.y_bstride, a.mean, b0, C, N, true);
        hipLaunchKernelGGL((big_matvec_kernel<false>), dim3(nmat), dim3(256), N * sizeof(float), st, Vm, wv, al, a.Y, a.y_bstride, a.mean, b0, C, N, false);
        hipLaunchKernelGGL(big_finish_kernel, dim3(nmat), dim3(256), 0, st, a, Lm, Vm, wv, al, jit, info_m, b0);
        if (a.flags & DKT_MLL_WANT_CHOL) hipLaunchKernelGGL(big_chol_out_kernel, dim3(32, nmat), dim3(256), 0, st, a, Lm, info_m, b0);
        if (want_grad) {
            // ---- K^-1 = V V^T, lower block triangle: row block i against the row blocks j <= i, all at once; the column blocks
            //      k >= i are contiguous: one GEMM per i with N = off(i) + sz(i) columns and K = N - off(i) ----
            for (int i = 0; i < nbk; ++i)
                gemm(st, nmat, false, true, sz(i), off(i) + sz(i), N - off(i), 1.f, blk(Vm, i, i), N, nn, blk(Vm, 0, i), N, nn, 0.f, blk(Kinv, i, 0), N, nn);
            hipLaunchKernelGGL(big_w_kernel, dim3(32, bcnt), dim3(256), 0, st, a, Kinv, al, info_m, b0, nb);
        }
        // ---- fix-up: episodes with a failed matrix (info != 0 in the outputs just written) are redone by the generic kernel with
        //      the jitter ladder 1e-6, 1e-5, 1e-4 per matrix; its global working matrices reuse the Kw / Lm region ----
        {
            MllArgs f = a;
            f.only_failed = a.info;
            dkt_mll_generic_global_launch(f, b0, bcnt, Kw, st);
        }
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
