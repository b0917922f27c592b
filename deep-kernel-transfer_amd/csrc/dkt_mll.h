// Shared between dkt_mll.hip (generic LDS/global path) and dkt_mll_reg.hip (register-resident path).
#pragma once
#include "dkt_common.h"
#include "../../include/dkt_abi.h"

struct MllArgs {
    const float* E;
    const float* Y;
    long y_bstride;
    const float* sv;
    const float* mean;
    const float* noise;
    const float* cls_weight;
    float* logp;
    float* alpha;
    float* L;
    float* W;
    float* dsv;
    float* dmean;
    float* dnoise;
    float* jitter_used;
    int32_t* info;
    float* ws;
    int B, C, N, LD;
    float jitter0;
    int max_tries;
    unsigned flags;
};

constexpr float DKT_HALF_LOG_2PI = 0.91893853320467274178f;

// Register-resident path: handles N + 1 <= 128.  Returns false when N is out of range.
bool dkt_mll_reg_launch(const MllArgs& a, hipStream_t st);
// Blocked path (panel sweep + MFMA trailing updates): N + 1 <= 128.
bool dkt_mll_blk_launch(const MllArgs& a, hipStream_t st);
// Wave-per-episode path (no barriers): N + 1 in (104, 112].  Returns false when N is out of range / disabled.
bool dkt_mll_wave_launch(const MllArgs& a, hipStream_t st);

// Batched factorisation + inversion of nb x nb diagonal blocks (nb <= 127) with the register-resident sweep (dkt_mll_reg.hip).
void dkt_chol_inv_block_launch(const float* A, int lda, long sA, float* L, int ldl, long sL, float* U, int ldu, long sU, int nb,
                               int pivot_base, int32_t* info, int nmat, hipStream_t st);
// Blocked path for N > 127 (dkt_mll_big.hip).
size_t dkt_mll_big_workspace_bytes(int B, int C, int N);
int dkt_mll_big_launch(const MllArgs& a, void* workspace, size_t ws_bytes, hipStream_t st);
