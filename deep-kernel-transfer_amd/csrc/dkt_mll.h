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
