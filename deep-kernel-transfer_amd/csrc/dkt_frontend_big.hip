// dkt_frontend_big.hip -- the front end of a training episode with MORE than 128 rows (the 20-way shapes of train.py:132-133: N = 420): train-mode
// BatchNorm1d + F.normalize in front of the large-N Gram kernels, and the way back, as three streaming kernels.
//
// Replaces bn_out in train mode + F.normalize (reference methods/DKT.py:48, 141-142) and autograd through both (loss.backward(), DKT.py:163) where the
// episode-resident fused kernels of dkt_frontend.hip (N <= 128: the normalised features never leave the chip) do not apply: the large-N Gram kernels need
// unit rows as their INPUT (they scale-split them to f16), so Zn is written once -- but by ONE kernel instead of the seven element-wise / reduction kernels
// torch runs for the same two modules (9.4 ms per 1024 episodes of 420 x 512 features, a third of the whole step: profiles/r04/v13_front_end_large_n.log).
//
//   forward : dkt_bn_stats_f32 (column statistics, folded into a / s)  ->  dkt_affine_normalize_f32:  y = a x + s,  rn = 1 / max(|y|, 1e-12),  Zn = y rn
//   backward: dkt_normalize_bn_bwd_f32:  t_i = Zn_i . dZn_i  (row kernel)
//             dY = rn (dZn - Zn t);  train mode: dgamma = sum_i dY xhat, dbeta = sum_i dY, dX = a (dY - dbeta / N - xhat dgamma / N)   (column kernel:
//             a workgroup per (episode, 32-feature slab) keeps the slab's dY in LDS between the two passes over the rows)
#include "dkt_common.h"
#include "../../include/dkt_abi.h"

namespace {

// one wave per row; up to D = 2048 the row's y = a x + s stays in the wave's registers between the norm and the store (X is read once), beyond that the row is
// read a second time (an L1 / L2 hit)
__global__ __launch_bounds__(256) void affine_normalize_kernel(const float* __restrict__ X, const float* __restrict__ A, const float* __restrict__ S, long ab_bstride,
                                                               float* __restrict__ Zn, float* __restrict__ rnorm, long rows, int N, int D) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const long b = row / N;
    const float4* x = reinterpret_cast<const float4*>(X + row * D);
    const float4* a = reinterpret_cast<const float4*>(A + b * ab_bstride);
    const float4* s = reinterpret_cast<const float4*>(S + b * ab_bstride);
    float4* z = reinterpret_cast<float4*>(Zn + row * D);
    const int nv = D >> 2;
    float ss = 0.f;
    if (nv <= 512) {
        float4 y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = lane + 64 * k;
            y[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < nv) {
                const float4 xv = x[v], av = a[v], sv = s[v];
                y[k] = make_float4(__builtin_fmaf(av.x, xv.x, sv.x), __builtin_fmaf(av.y, xv.y, sv.y), __builtin_fmaf(av.z, xv.z, sv.z), __builtin_fmaf(av.w, xv.w, sv.w));
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {      // (same order of the squares as the streaming form below: lane-wise v = lane, lane + 64, ...)
            ss = __builtin_fmaf(y[k].x, y[k].x, ss); ss = __builtin_fmaf(y[k].y, y[k].y, ss); ss = __builtin_fmaf(y[k].z, y[k].z, ss); ss = __builtin_fmaf(y[k].w, y[k].w, ss);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, DKT_WAVE);
        const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);        // F.normalize: x / max(|x|_2, eps)
        if (lane == 0) rnorm[row] = rn;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int v = lane + 64 * k;
            if (v < nv) z[v] = make_float4(y[k].x * rn, y[k].y * rn, y[k].z * rn, y[k].w * rn);
        }
        return;
    }
    for (int v = lane; v < nv; v += 64) {
        const float4 xv = x[v], av = a[v], sv = s[v];
        const float y0 = __builtin_fmaf(av.x, xv.x, sv.x), y1 = __builtin_fmaf(av.y, xv.y, sv.y), y2 = __builtin_fmaf(av.z, xv.z, sv.z), y3 = __builtin_fmaf(av.w, xv.w, sv.w);
        ss = __builtin_fmaf(y0, y0, ss); ss = __builtin_fmaf(y1, y1, ss); ss = __builtin_fmaf(y2, y2, ss); ss = __builtin_fmaf(y3, y3, ss);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, DKT_WAVE);
    const float rn = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    if (lane == 0) rnorm[row] = rn;
    for (int v = lane; v < nv; v += 64) {                     // (the second read of the row is an L1 / L2 hit)
        const float4 xv = x[v], av = a[v], sv = s[v];
        z[v] = make_float4(__builtin_fmaf(av.x, xv.x, sv.x) * rn, __builtin_fmaf(av.y, xv.y, sv.y) * rn, __builtin_fmaf(av.z, xv.z, sv.z) * rn,
                           __builtin_fmaf(av.w, xv.w, sv.w) * rn);
    }
}

// t[row] = Zn[row] . dZn[row]; one wave per row
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ Zn, const float* __restrict__ dZn, float* __restrict__ t, long rows, int D) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float4* z = reinterpret_cast<const float4*>(Zn + row * D);
    const float4* g = reinterpret_cast<const float4*>(dZn + row * D);
    float acc = 0.f;
    for (int v = lane; v < (D >> 2); v += 64) {
        const float4 zv = z[v], gv = g[v];
        acc = __builtin_fmaf(zv.x, gv.x, acc); acc = __builtin_fmaf(zv.y, gv.y, acc); acc = __builtin_fmaf(zv.z, gv.z, acc); acc = __builtin_fmaf(zv.w, gv.w, acc);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, DKT_WAVE);
    if (lane == 0) t[row] = acc;
}

// workgroup = (episode b, 32-feature slab): thread (r = tid >> 3, c4 = tid & 7) walks the rows r, r + 32, ... of its 4 features.
// TRAIN: first pass dY -> LDS + column sums, second pass dX from LDS (x re-read: an L2 hit); otherwise (eval-mode statistics / no bn_out) dX = a dY at once.
template <bool TRAIN>
__global__ __launch_bounds__(256) void normalize_bn_bwd_cols_kernel(const float* __restrict__ dZn, const float* __restrict__ Zn, const float* __restrict__ X,
                                                                    const float* __restrict__ A, long a_bstride, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ rnorm, const float* __restrict__ t,
                                                                    float* __restrict__ dX, float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                                    int N, int D, int nslab) {
    extern __shared__ __attribute__((aligned(16))) float dy_s[];            // TRAIN: [N][32] | red[2][32][8 x 4]
    const int b = blockIdx.x / nslab, sl = blockIdx.x % nslab;
    const int tid = threadIdx.x, r = tid >> 3, c4 = tid & 7;
    const int d = 32 * sl + 4 * c4;
    const bool dok = d < D;                                                 // (D % 4 == 0: a float4 is inside the row or wholly beyond it)
    const size_t base = (size_t)b * N * D + d;
    const float4 av = dok ? *reinterpret_cast<const float4*>(A + (size_t)b * a_bstride + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = mu;
    if (TRAIN && dok) {
        mu = *reinterpret_cast<const float4*>(mean + (size_t)b * D + d);
        rs = *reinterpret_cast<const float4*>(rstd + (size_t)b * D + d);
    }
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;                   // column sums of dY and dY xhat over this thread's rows
    // four rows per trip, every load of the trip issued before the first use (round 6: one row per trip left each of the 16 waves of a CU waiting for its own
    // three loads, trip after trip -- 0.40 of the HBM roofline at the 20-way shape); rows past N are clamped to the last row and weighted out
    for (int i0 = r; i0 < N; i0 += 128) {
        float4 g[4], z[4], x[4];
        float ti[4], rn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + 32 * u, N - 1);
            const size_t o = base + (size_t)i * D;
            g[u] = z[u] = x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dok) {
                g[u] = *reinterpret_cast<const float4*>(dZn + o);
                z[u] = *reinterpret_cast<const float4*>(Zn + o);
                if (TRAIN) x[u] = *reinterpret_cast<const float4*>(X + o);
            }
            ti[u] = t[(size_t)b * N + i];
            rn[u] = rnorm[(size_t)b * N + i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 32 * u;
            if (i < N) {
                float4 dy = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dok) {
                    dy = make_float4(rn[u] * __builtin_fmaf(-z[u].x, ti[u], g[u].x), rn[u] * __builtin_fmaf(-z[u].y, ti[u], g[u].y),
                                     rn[u] * __builtin_fmaf(-z[u].z, ti[u], g[u].z), rn[u] * __builtin_fmaf(-z[u].w, ti[u], g[u].w));
                    if (TRAIN) {
                        sb.x += dy.x; sb.y += dy.y; sb.z += dy.z; sb.w += dy.w;
                        sg.x = __builtin_fmaf(dy.x, (x[u].x - mu.x) * rs.x, sg.x); sg.y = __builtin_fmaf(dy.y, (x[u].y - mu.y) * rs.y, sg.y);
                        sg.z = __builtin_fmaf(dy.z, (x[u].z - mu.z) * rs.z, sg.z); sg.w = __builtin_fmaf(dy.w, (x[u].w - mu.w) * rs.w, sg.w);
                    } else {
                        *reinterpret_cast<float4*>(dX + base + (size_t)i * D) = make_float4(av.x * dy.x, av.y * dy.y, av.z * dy.z, av.w * dy.w);
                    }
                }
                if (TRAIN) *reinterpret_cast<float4*>(dy_s + (size_t)i * 32 + 4 * c4) = dy;
            }
        }
    }
    if (!TRAIN) return;
    // the 32 row-threads of a feature quad: fixed-order tree over r in LDS (deterministic)
    float* red = dy_s + (size_t)N * 32;                                     // [32 rows r][8 c4][8]
    *reinterpret_cast<float4*>(red + (r * 8 + c4) * 8) = sb;
    *reinterpret_cast<float4*>(red + (r * 8 + c4) * 8 + 4) = sg;
    __syncthreads();
    for (int step = 16; step >= 1; step >>= 1) {
        if (r < step) {
            float* p = red + (r * 8 + c4) * 8;
            const float* qv = red + ((r + step) * 8 + c4) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) p[e] += qv[e];
        }
        __syncthreads();
    }
    const float4 tb = *reinterpret_cast<const float4*>(red + c4 * 8), tg = *reinterpret_cast<const float4*>(red + c4 * 8 + 4);
    if (r == 0 && dok) {
        *reinterpret_cast<float4*>(dbeta_part + (size_t)b * D + d) = tb;
        *reinterpret_cast<float4*>(dgamma_part + (size_t)b * D + d) = tg;
    }
    if (!dok) return;
    const float inv_n = 1.0f / (float)N;
    for (int i0 = r; i0 < N; i0 += 128) {
        float4 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4*>(X + base + (size_t)min(i0 + 32 * u, N - 1) * D);     // (second read of the slab: an L2 / MALL hit)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 32 * u;
            if (i < N) {
                const float4 dy = *reinterpret_cast<const float4*>(dy_s + (size_t)i * 32 + 4 * c4);
                float4 out;
                out.x = av.x * (dy.x - inv_n * (tb.x + (x[u].x - mu.x) * rs.x * tg.x));
                out.y = av.y * (dy.y - inv_n * (tb.y + (x[u].y - mu.y) * rs.y * tg.y));
                out.z = av.z * (dy.z - inv_n * (tb.z + (x[u].z - mu.z) * rs.z * tg.z));
                out.w = av.w * (dy.w - inv_n * (tb.w + (x[u].w - mu.w) * rs.w * tg.w));
                *reinterpret_cast<float4*>(dX + base + (size_t)i * D) = out;
            }
        }
    }
}

}  // namespace

extern "C" int dkt_affine_normalize_f32(const float* X, const float* a, const float* s, long ab_bstride, float* Zn, float* rnorm, int B, int N, int D, void* stream) {
    if (!X || !a || !s || !Zn || !rnorm || B <= 0 || N <= 0 || D <= 0) return DKT_ERR_BAD_ARG;
    if ((D & 3) || ((uintptr_t)X & 15) || ((uintptr_t)Zn & 15) || ((uintptr_t)a & 15) || ((uintptr_t)s & 15) || (ab_bstride & 3) || ab_bstride < 0) return DKT_ERR_BAD_ARG;
    const long rows = (long)B * N;
    if ((rows + 3) / 4 > 0x7fffffffL) return DKT_ERR_TOO_LARGE;
    hipLaunchKernelGGL(affine_normalize_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), dkt_lds_pad("DKT_PAD_AFFNORM"), (hipStream_t)stream, X, a, s, ab_bstride, Zn, rnorm, rows, N, D);
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}

extern "C" int dkt_normalize_bn_bwd_f32(const float* dZn, const float* Zn, const float* X, const float* a, long a_bstride, const float* mean, const float* rstd,
                                        const float* rnorm, float* dX, float* dgamma_part, float* dbeta_part, float* rowdot_ws, int B, int N, int D, void* stream) {
    if (!dZn || !Zn || !a || !rnorm || !dX || !rowdot_ws || B <= 0 || N <= 0 || D <= 0) return DKT_ERR_BAD_ARG;
    const bool train = mean != nullptr;
    if (train && (!X || !rstd || !dgamma_part || !dbeta_part)) return DKT_ERR_BAD_ARG;
    if ((D & 3) || ((uintptr_t)dZn & 15) || ((uintptr_t)Zn & 15) || ((uintptr_t)X & 15) || ((uintptr_t)dX & 15) || ((uintptr_t)a & 15) || (a_bstride & 3) || a_bstride < 0)
        return DKT_ERR_BAD_ARG;
    if (train && (((uintptr_t)mean & 15) || ((uintptr_t)rstd & 15) || ((uintptr_t)dgamma_part & 15) || ((uintptr_t)dbeta_part & 15))) return DKT_ERR_BAD_ARG;
    if (N > 1024) return DKT_ERR_TOO_LARGE;                                  // the slab's dY lives in LDS: 128 B per row
    const long rows = (long)B * N;
    const int nslab = (D + 31) / 32;
    if ((rows + 3) / 4 > 0x7fffffffL || (long)B * nslab > 0x7fffffffL) return DKT_ERR_TOO_LARGE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), dkt_lds_pad("DKT_PAD_ROWDOT"), st, Zn, dZn, rowdot_ws, rows, D);
    if (train) {
        const size_t lds = ((size_t)N * 32 + 32 * 8 * 8) * sizeof(float);
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute((const void*)normalize_bn_bwd_cols_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((1024 * 32 + 32 * 8 * 8) * sizeof(float))) != hipSuccess)
            return DKT_ERR_LAUNCH;
        hipLaunchKernelGGL(normalize_bn_bwd_cols_kernel<true>, dim3((unsigned)(B * nslab)), dim3(256), lds, st, dZn, Zn, X, a, a_bstride, mean, rstd, rnorm, rowdot_ws, dX,
                           dgamma_part, dbeta_part, N, D, nslab);
    } else {
        hipLaunchKernelGGL(normalize_bn_bwd_cols_kernel<false>, dim3((unsigned)(B * nslab)), dim3(256), dkt_lds_pad("DKT_PAD_NBB"), st, dZn, Zn, X, a, a_bstride, mean, rstd, rnorm, rowdot_ws, dX,
                           dgamma_part, dbeta_part, N, D, nslab);
    }
    return hipGetLastError() == hipSuccess ? DKT_OK : DKT_ERR_LAUNCH;
}
