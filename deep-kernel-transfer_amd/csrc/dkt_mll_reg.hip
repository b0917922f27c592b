// dkt_mll_reg.hip -- batched factorisation + inversion of the diagonal blocks of the blocked path for N > 127 (dkt_mll_big.hip) with the register-resident
// sweep of dkt_reg_sweep.h.  (Rounds 1-3 this file also held mll_reg_kernel, the round-1 marginal-likelihood kernel for N <= 127 and later the
// DKT_MLL_FORCE_REG validation twin of the MFMA kernels: it moved to the measurement library, dkt_mll_reg_twin.hip, in round 4.)
#include "dkt_reg_sweep.h"

namespace {

// ---------------------------------------------------------------------------------------------
// chol_inv_block_kernel<NT>: the same sweep on a batch of nb x nb SPD blocks that live inside larger matrices (leading
// dimension ld): block A (lower triangle read) -> L (lower incl. diagonal) and U = L^-T (upper incl. diagonal = 1 / L_kk).
// Building block of the blocked path for N > 127 (dkt_mll_big.hip): with U at hand the panel step L_ij = A_ij U_jj is a GEMM.
// info[m] receives pivot_base + (index of the first non-positive pivot) + 1 unless an earlier block already failed.
template <int NT>
__global__ __launch_bounds__(256, NT <= 7 ? 4 : 2) void chol_inv_block_kernel(const float* __restrict__ A, int lda, long sA,
                                                                             float* __restrict__ L, int ldl, long sL,
                                                                             float* __restrict__ U, int ldu, long sU, int nb,
                                                                             int pivot_base, int32_t* __restrict__ info) {
    constexpr int NP = 16 * NT;
    __shared__ float colbuf[4 * NP];
    __shared__ float dgv[NP];
    const int m = blockIdx.x, tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const float* Ab = A + (size_t)m * sA;
    float* Lb = L + (size_t)m * sL;
    float* Ub = U + (size_t)m * sU;
    const int tyN = nb - 16 * (NT - 1);                  // valid columns of the last block column (row nb is an unused zero row)
    const bool col_ok = tx < tyN;
    f32x2 A2[(NT + 1) / 2][NT];
#pragma unroll
    for (int pi = 0; pi < NT; ++pi) {
#pragma unroll
        for (int ji = 0; ji < NT; ++ji) {
            const int p = ty + 16 * pi, j = tx + 16 * ji;
            float v = 0.f;
            if (pi >= ji && p >= j && p < nb && j < nb) v = Ab[(size_t)p * lda + j];
            AE(pi, ji) = v;
        }
    }
    RegCtx<NT> ctx;
    ctx.colbuf = colbuf; ctx.N = nb; ctx.tx = tx; ctx.ty = ty; ctx.tid = tid; ctx.col_ok = col_ok;
    __syncthreads();
    sweep_all<NT, 0>(A2, ctx);
    __builtin_amdgcn_s_setprio(0);
    if (ty == tx) {
#pragma unroll
        for (int ji = 0; ji < NT; ++ji) dgv[tx + 16 * ji] = AE(ji, ji);
    }
    __syncthreads();
    float rinvcol[NT];
    int bad = 0x7fffffff;
#pragma unroll
    for (int ji = NT - 1; ji >= 0; --ji) {
        const float dj = dgv[tx + 16 * ji];
        const bool valid = (ji < NT - 1) || col_ok;
        rinvcol[ji] = valid ? __builtin_amdgcn_rsqf(dj) : 1.0f;
        bad = (valid && !(dj > 0.f)) ? tx + 16 * ji + 1 : bad;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) bad = min(bad, __shfl_xor(bad, o, DKT_WAVE));
    if (tid == 0 && bad != 0x7fffffff && info[m] == 0) info[m] = pivot_base + bad;
#pragma unroll
    for (int pi = 0; pi < NT; ++pi) {
#pragma unroll
        for (int ji = 0; ji < NT; ++ji) {
            const int p = ty + 16 * pi, j = tx + 16 * ji;
            if (p < nb && j < nb) {
                const float v = AE(pi, ji) * rinvcol[ji];
                if (p > j) {
                    Lb[(size_t)p * ldl + j] = v;
                    Ub[(size_t)p * ldu + j] = 0.f;          // consumers multiply with the full U block: its lower part must be 0
                } else if (p < j) Ub[(size_t)p * ldu + j] = v;
                else {
                    Lb[(size_t)p * ldl + j] = 1.0f / rinvcol[ji];
                    Ub[(size_t)p * ldu + j] = rinvcol[ji];
                }
            }
        }
    }
}

}  // namespace

void dkt_chol_inv_block_launch(const float* A, int lda, long sA, float* L, int ldl, long sL, float* U, int ldu, long sU, int nb,
                               int pivot_base, int32_t* info, int nmat, hipStream_t st) {
    switch ((nb + 1 + 15) / 16) {
        case 1: hipLaunchKernelGGL((chol_inv_block_kernel<1>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 2: hipLaunchKernelGGL((chol_inv_block_kernel<2>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 3: hipLaunchKernelGGL((chol_inv_block_kernel<3>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 4: hipLaunchKernelGGL((chol_inv_block_kernel<4>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        // blocks of the blocked path are nb = 64 rows (the last one of a matrix fewer): NT <= 5; larger blocks only with the measurement switch DKT_BIG_NB
#ifdef DKT_TWINS
        case 5: hipLaunchKernelGGL((chol_inv_block_kernel<5>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 6: hipLaunchKernelGGL((chol_inv_block_kernel<6>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        case 7: hipLaunchKernelGGL((chol_inv_block_kernel<7>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
        default: hipLaunchKernelGGL((chol_inv_block_kernel<8>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
#else
        default: hipLaunchKernelGGL((chol_inv_block_kernel<5>), dim3(nmat), dim3(256), 0, st, A, lda, sA, L, ldl, sL, U, ldu, sU, nb, pivot_base, info); break;
#endif
    }
}
