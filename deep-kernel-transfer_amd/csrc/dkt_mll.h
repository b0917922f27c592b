// Shared between dkt_mll.hip (dispatch + generic LDS/global path), dkt_mll_mfma.hip, dkt_mll_reg.hip and dkt_mll_big.hip.
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
    const int32_t* only_failed;   // generic kernel as the fix-up pass of the blocked path: redo only episodes with a non-zero entry here
    int b0;                       // first episode of the launch (workgroup x handles episode b0 + x; the workspace is indexed by x)
    int B, C, N, LD;
    float jitter0;
    int max_tries;
    unsigned flags;
    int p2_guard;                 // wave-per-episode kernel: binades of head room of the f16 scale of M over the diagonal tiles (DKT_MLL_P2_GUARD, default 1)
    float kappa_max = 0.f;        // f16-split kernels (dkt_mll_h2.hip): > 0 = a class whose a-priori bound 1 + sv trace(E) / noise exceeds it (and whose factorisation succeeded)
                                  // leaves with info = -1 for the generic kernel's fix-up launch (dkt_mll.hip, mll_kappa_fixup); 0 = no test
};

constexpr float DKT_HALF_LOG_2PI = 0.91893853320467274178f;

// Wave-per-class-matrix path on the f16 matrix pipe (dkt_mll_h2.hip): the default for N + 1 <= 128 unless the Cholesky factors are
// requested.  Returns false when it does not apply.
bool dkt_mll_h2_launch(const MllArgs& a, hipStream_t st);
// The same structure on the exact-fp32 matrix instruction (dkt_mll_mfma.hip): DKT_MLL_FORCE_F32MFMA twin, and DKT_MLL_WANT_CHOL.
// Returns false when N is out of range.
bool dkt_mll_mfma_launch(const MllArgs& a, hipStream_t st);
// Register-resident sweep (dkt_mll_reg_twin.hip; the round-1 default, a validation twin in libdkt_diag.so since round 4): N + 1 <= 128.
bool dkt_mll_reg_launch(const MllArgs& a, hipStream_t st);

// Batched factorisation + inversion of nb x nb diagonal blocks (nb <= 127) with the register-resident sweep (dkt_mll_reg.hip).
void dkt_chol_inv_block_launch(const float* A, int lda, long sA, float* L, int ldl, long sL, float* U, int ldu, long sU, int nb,
                               int pivot_base, int32_t* info, int nmat, hipStream_t st);
// Blocked path for N > 127 (dkt_mll_big.hip).
size_t dkt_mll_big_workspace_bytes(int B, int C, int N);
int dkt_mll_big_launch(const MllArgs& a, void* workspace, size_t ws_bytes, hipStream_t st);
// Tile-array path for N > 127 (dkt_mll_tiled.hip; the default there unless the Cholesky factors are requested).
bool dkt_mll_tiled_supports(int N, unsigned flags, int C = 1);
size_t dkt_mll_tiled_workspace_bytes(int B, int C, int N);
size_t dkt_mll_tiled_workspace_bytes_form(int B, int C, int N, bool per_class);
int dkt_mll_tiled_launch(const MllArgs& a, void* workspace, size_t ws_bytes, hipStream_t st);
// Shared-E band path (dkt_mll_band.hip): ONE orthogonal reduction per episode instead of C factorisations (128 <= N <= 432, C <= 32; the default from 12 classes and 192 episodes per call).
bool dkt_mll_band_supports(int N, unsigned flags, int C);
bool dkt_mll_band_applies(int B, int C, int N, unsigned flags);      // the default dispatch window (C >= 12, B >= 192) or DKT_MLL_FORCE_BAND
size_t dkt_mll_band_workspace_bytes(int B, int C, int N);
int dkt_mll_band_launch(const MllArgs& a, void* workspace, size_t ws_bytes, hipStream_t st);
// Generic kernel (dkt_mll.hip) over episodes [b0, b0 + count), global working matrices in `ws`; with a.only_failed set it recomputes
// -- with the full jitter-retry ladder -- only the episodes the blocked path reported as failed.  No host synchronisation.
void dkt_mll_generic_global_launch(MllArgs a, int b0, int count, float* ws, hipStream_t st);
size_t dkt_mll_generic_global_floats(int count, int N);
