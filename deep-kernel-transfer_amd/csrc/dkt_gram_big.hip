// dkt_gram_big.hip -- Gram backward for 128 < N <= 448 (the 20-way shapes) on the f16 MFMA pipe:  dZ[b] = s_b (W[b] + W[b]^T) Z[b]
// for unit-norm rows of Z (DKT_GRAM_UNIT_ROWS) and a W the caller declares symmetric (DKT_GRAM_W_SYMMETRIC: dkt_mll_f32 writes
// W[b] bitwise symmetric), so that the A operand 2 s W is built from ROW reads only.
//
// Replaces autograd through matmul(Z, Z^T) in loss.backward() (reference methods/DKT.py:163) at the sizes of train.py:132-133
// (n_way = 20), where the episode-resident kernel of dkt_gram_ep.hip (N <= 128: one workgroup holds every row's A fragments) no
// longer applies and the generic fp32 kernel of dkt_gram.hip ran at 0.08 of the HBM roof.
//
// Workgroup = 4 waves = 64 output rows of one episode; wave w keeps the A fragments of its 16 rows for the WHOLE contraction range
// (KS k32-slices x 2 f16 planes: 8 KS VGPRs) in registers, every row scaled by its own power of two (row maximum -> [2^14, 2^15))
// before the 2-way f16 split (dkt_split.h) -- exact, undone per output row in the epilogue.  Z is streamed in slabs of 32 features:
// all N rows of the slab are split (scaled by 2^15) and transposed into a [feature][row] f16 image in LDS (two planes; the same
// staging as gram_bwd_ep_f16x2_kernel), from which a lane's 8 consecutive-row B values are one ds_read_b128.  The row blocks of an
// episode re-read the same Z slabs: the workgroup -> (episode, row block) map keeps them on ONE XCD (workgroup id mod 8 = XCD), so
// the re-reads are served by that XCD's L2.
#include <type_traits>
#include "dkt_split.h"
#include "../../include/dkt_abi.h"

bool dkt_gram_split_enabled();           // dkt_gram_ep.hip: DKT_GRAM_SPLIT
#define DKT_F16_UNSCALE_BIG (1.f / (32768.f * 32768.f))

namespace {

// NW = 4: round 2's kernel (64 output rows per workgroup, one LDS image, two barriers per slab, 2 workgroups per CU).
// NW = 8 (round 4, default; DKT_GRAM_BWD_ROWS8=0 restores NW = 4): 128 output rows per workgroup -- the slab is split / transposed / staged 4 instead
// of 7 times per N = 420 episode (the NW = 4 kernel issues 4.8 VALU instructions per MFMA, the matrix pipe 22 % busy, and half of its LDS cycles are bank
// conflicts of the transposing stores: profiles/r04/v0_gram_big_pmc.txt) --, two LDS images (one barrier per slab), and the features of a slab permuted over
// the image rows so that the 8-byte transposing stores of a wave fall into distinct banks (image row = 16 (t & 1) + d4 + 8 (t >> 1) for feature 4 d4 + t:
// the 8 rows a 16-lane store group touches are CONSECUTIVE, 40 banks apart mod 64, instead of every second row, 16 banks apart).
template <int KS, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void gram_bwd_rows_f16x2_kernel(const float* __restrict__ W, const float* __restrict__ Z,
                                                                    float* __restrict__ dZ, int B, int N, int D,
                                                                    const float* __restrict__ ep_scale, int nrb) {
    constexpr int KP = 32 * KS;
    constexpr int BD = 32;
    constexpr int SU = (KP / 8) + ((KP / 8) % 4 == 2 ? 0 : (6 - (KP / 8) % 4) % 4);
    constexpr int RS = 8 * SU;                           // f16 per LDS row (one feature, all rows j), 16-byte units == 2 mod 4
    constexpr int PLANE = BD * RS;
    constexpr int RPP = 32 * NW;                         // rows staged per pass (128 / 256)
    constexpr int NPASS = (KP + RPP - 1) / RPP;
    constexpr int NBUF = NW == 4 ? 1 : 2;
    static_assert(SU % 4 == 2 && RS >= KP, "LDS row stride");
    __shared__ __attribute__((aligned(16))) _Float16 zt[NBUF][2 * PLANE];
    __shared__ float rowinv[16 * NW];

    // workgroup -> (episode, row block): consecutive workgroup ids go to consecutive XCDs, so the row blocks of one episode are
    // given ids that are 8 apart
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int b = (slot / nrb) * 8 + xcd, rb = slot % nrb;
    if (b >= B) return;
    const int r0 = 16 * NW * rb;
    const float* Wb = W + (size_t)b * N * N;
    const float* Zb = Z + (size_t)b * N * D;
    float* dZb = dZ + (size_t)b * N * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const float s2 = 2.0f * (ep_scale ? ep_scale[b] : 1.0f);

    // ---- Z staging: thread = 4 rows (4 jg .. 4 jg + 3, + 128 per pass) x 4 features (4 d4 .. 4 d4 + 3) ----
    const int d4 = tid & 7, jg = tid >> 3;
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Zb), 0, N * D * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t dzr = __builtin_amdgcn_make_buffer_rsrc(dZb, 0, N * D * 4, 0x00020000);
    // The per-thread offsets are loop-invariant (round 5): a feature past D in a ragged last slab is NOT masked -- it reads the next row's first features
    // (unit-row values; behind the last row: 0 through the descriptor) into image columns whose outputs the dZ store drops (a column of dZ depends on
    // that column of Z only).  With the mask the offsets were recomputed per slab INTO the registers the loads return to, and the `vmcnt` waits hipcc
    // put in front of those writes (it cannot prove across the back edge that the previous loads have landed) drained the slab's dZ stores every trip:
    // the 12 - 16 % "issue of the next slab's loads" of profiles/r04/v12_gram_bwd_phase_clocks.log.
    auto gload = [&](float4 (&rg)[NPASS][4], int d0) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = RPP * p + 4 * jg + rr;
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(zr, (row < N) ? (row * D + 4 * d4) * 4 : 0x7ffffff0, d0 * 4, 0);
                rg[p][rr] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
    };
    // feature d = 4 d4 + t of the slab goes to LDS row 16 (d & 1) + (d >> 1): output tile tt = d & 1 holds the features {2 r + tt},
    // so the two accumulators of a lane are 2 consecutive features (8-byte stores of dZ)
    auto lstore = [&](const float4 (&rg)[NPASS][4], const int buf) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if (RPP * p + 4 * jg < KP) {
                const float x[4][4] = {{rg[p][0].x, rg[p][1].x, rg[p][2].x, rg[p][3].x}, {rg[p][0].y, rg[p][1].y, rg[p][2].y, rg[p][3].y},
                                       {rg[p][0].z, rg[p][1].z, rg[p][2].z, rg[p][3].z}, {rg[p][0].w, rg[p][1].w, rg[p][2].w, rg[p][3].w}};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f16x4 h, m;
                    split2h(make_float4(x[t][0], x[t][1], x[t][2], x[t][3]), 32768.f, h, m);
                    const int lrow = NW == 4 ? 16 * (t & 1) + 2 * d4 + (t >> 1) : 16 * (t & 1) + d4 + 8 * (t >> 1);
                    _Float16* dst = zt[buf] + lrow * RS + RPP * p + 4 * jg;
                    *reinterpret_cast<f16x4*>(dst) = h;
                    *reinterpret_cast<f16x4*>(dst + PLANE) = m;
                }
            }
        }
    };
    const int nslab = (D + BD - 1) / BD;
    float4 rg[NPASS][4];

    // ---- A fragments: row r0 + 16 wave + r16, slot e of slice ks on lane (r16, q) is k = 32 ks + 8 q + e; value 2 s W[row][k] ----
    f16x8 ah[KS], am[KS];
    {
        const int row = r0 + 16 * wave + r16;
        const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wb), 0, N * N * 4, 0x00020000);
        // BRANCH-FREE (round 4): a 16-byte load that starts inside the row is issued whole -- what it reads past the row end is the next row (or, for the
        // episode's last row, beyond the buffer: 0) and is masked after the load.  With the element-by-element tail behind a lane-dependent branch the
        // compiler waited for every load before the next one was issued: 56 serialized round trips per wave, 0.3 of the kernel's 1.26 ms at N = 420
        // (profiles/r04/v11_gram_bwd_variants.log).
        auto wload = [&](float (&v)[8], int ks) {
            const int k = 32 * ks + 8 * q;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int kk = k + 4 * hh;
                const auto u = __builtin_amdgcn_raw_buffer_load_b128(wr, (row < N && kk < N) ? (row * N + kk) * 4 : 0x7ffffff0, 0, 0);
                // (the launcher picks KS with 32 (KS - 2) < N: only the last two slices can reach past the row end -- masks for all of them cost 150 SGPR spills)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * hh + e] = (ks < KS - 2 || kk + e < N) ? s2 * __uint_as_float(u[e]) : 0.f;
            }
        };
        float rmax = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float v[8];
            wload(v, ks);
#pragma unroll
            for (int e = 0; e < 8; ++e) rmax = fmaxf(rmax, fabsf(v[e]));
        }
        rmax = fmaxf(rmax, __shfl_xor(rmax, 16, DKT_WAVE));
        rmax = fmaxf(rmax, __shfl_xor(rmax, 32, DKT_WAVE));
        // power-of-two row scale: row maximum -> [2^14, 2^15); clamped so that its inverse (times 2^-15) stays normal
        const int eb = (int)((__float_as_uint(rmax) >> 23) & 0xffu);
        const int sexp = min(268 - eb, 237);
        const float rscale = __uint_as_float((unsigned)sexp << 23);
        if (q == 0) rowinv[16 * wave + r16] = __uint_as_float((unsigned)(254 - sexp - 15) << 23);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            float v[8];
            wload(v, ks);                                // second pass: served by L2
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xs = v[e] * rscale;
                const _Float16 hi = (_Float16)xs;
                ah[ks][e] = hi;
                am[ks][e] = (_Float16)(xs - (float)hi);
            }
        }
    }
#ifdef DKT_GBW_CLOCKS       // measurement build (tools/gbw_phase_clocks.py): shader clocks of wave 0 per phase, summed over the slabs, left in the first row of the block
    unsigned long long ck[6] = {0, 0, 0, 0, 0, 0}, tk0 = __builtin_amdgcn_s_memtime();
    const unsigned long long tk_start = tk0, rt_start = __builtin_amdgcn_s_memrealtime();      // (the second: a constant 100-MHz counter -> the shader clock of this run)
#define GBW_CLK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long tk1 = __builtin_amdgcn_s_memtime(); ck[i] += tk1 - tk0; tk0 = tk1; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GBW_CLK(i) do { } while (0)
#endif
    gload(rg, 0);
    lstore(rg, 0);
    __syncthreads();
    GBW_CLK(0);                                              // prologue (A fragments) + first slab staged
    for (int sl = 0; sl < nslab; ++sl) {
        if (sl + 1 < nslab) gload(rg, (sl + 1) * BD);
        GBW_CLK(1);                                          // issue of the next slab's loads
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const _Float16* base = zt[NBUF == 2 ? (sl & 1) : 0] + r16 * RS + 8 * q;
        if (r0 + 16 * wave < N) {                            // (a wave of the episode's last block whose 16 rows lie past N only stages)
            if constexpr (NW == 8) {
                // B fragments (slice i >> 1, feature tile i & 1) read three steps (of 3 MFMAs) ahead into a ring of four register sets: left to itself the
                // compiler issues a step's two ds_read_b128 right in front of its MFMAs
                f16x8 rbh[4], rbm[4];
                auto fload = [&](const int i, const int slot) {
                    const _Float16* p = base + 16 * (i & 1) * RS + 32 * (i >> 1);
                    rbh[slot] = *reinterpret_cast<const f16x8*>(p);
                    rbm[slot] = *reinterpret_cast<const f16x8*>(p + PLANE);
                };
                fload(0, 0);
                fload(1, 1);
                fload(2, 2);
#pragma unroll
                for (int i = 0; i < 2 * KS; ++i) {
                    if (i + 3 < 2 * KS) fload(i + 3, (i + 3) % 4);
                    const int ks = i >> 1, tt = i & 1;
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], rbm[i % 4], acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am[ks], rbh[i % 4], acc[tt], 0, 0, 0);
                    acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], rbh[i % 4], acc[tt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const _Float16* p = base + 16 * tt * RS + 32 * ks;
                        const f16x8 bh = *reinterpret_cast<const f16x8*>(p);
                        const f16x8 bm = *reinterpret_cast<const f16x8*>(p + PLANE);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bm, acc[tt], 0, 0, 0);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am[ks], bh, acc[tt], 0, 0, 0);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bh, acc[tt], 0, 0, 0);
                    }
                }
            }
        }
#ifdef DKT_GBW_CLOCKS
        asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[1]));  // the products are complete here
#endif
        GBW_CLK(2);                                          // products
        const int d = sl * BD + (NW == 4 ? 2 * r16 : 4 * (r16 & 7) + 2 * (r16 >> 3));
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int lr = 16 * wave + 4 * q + reg, row = r0 + lr;
            const float u = rowinv[lr];
#ifdef DKT_GRAM_BWD_BRANCHY_STORES        // the round-4 form (A/B builds)
            if (row < N && d < D) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 o = {acc[0][reg] * u, acc[1][reg] * u};
                __builtin_nontemporal_store(o, reinterpret_cast<f32x2*>(dZb + (size_t)row * D + d));
            }
#else
            // Branch-free (round 5, as in gram_bwd_ep_f16x2_kernel): behind a branch the compiler counts no store as outstanding and waits `vmcnt(0)` for the
            // last register of the next slab -- which drains THIS slab's stores (their write acknowledgements) before the staging can go on, with one
            // workgroup per CU and nothing else to run.  Rows past N fall behind the descriptor's end by themselves; a feature past D needs the mask.
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 o = {__float_as_uint(acc[0][reg] * u), __float_as_uint(acc[1][reg] * u)};
            __builtin_amdgcn_raw_buffer_store_b64(o, dzr, (d < D) ? (row * D + d) * 4 : 0x7ffffff0, 0, 2);
#endif
        }
        GBW_CLK(3);                                          // dZ stores
        if constexpr (NBUF == 1) __syncthreads();
        if (sl + 1 < nslab) lstore(rg, NBUF == 2 ? ((sl + 1) & 1) : 0);
        GBW_CLK(4);                                          // wait for the next slab's loads, split, transposing LDS stores
        __syncthreads();
        GBW_CLK(5);                                          // barrier
    }
#ifdef DKT_GBW_CLOCKS
    if (tid == 0 && D >= 8) {
        __builtin_amdgcn_s_waitcnt(0);
        for (int i = 0; i < 6; ++i) dZb[(size_t)r0 * D + i] = (float)ck[i];
        dZb[(size_t)r0 * D + 6] = (float)(__builtin_amdgcn_s_memtime() - tk_start);
        dZb[(size_t)r0 * D + 7] = (float)(__builtin_amdgcn_s_memrealtime() - rt_start);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// Symmetric linear Gram for N > 128: 64 x 64 output tiles like the generic fp32 kernel of dkt_gram.hip (which is bound by the fp32
// MFMA pipe there: 127-132 TF), with the operands split while they are staged (dkt_split.h) -- SPL = 2: two scaled f16 planes for
// unit-norm rows (DKT_KERNEL_LINEAR_UNIT), 3 x v_mfma_f32_16x16x32_f16 per 32-wide slice and tile; SPL = 3: the exact 3-way bf16
// split for operands of any range, 6 x v_mfma_f32_16x16x32_bf16 -- instead of 8 x v_mfma_f32_16x16x4_f32; two-level accumulation as in
// the episode-resident kernels.  Only the lower tiles are computed and mirrored; the tiles of an episode are mapped to one XCD (they
// re-read the same rows of Z).
template <int SPL>
__global__ __launch_bounds__(256, SPL == 2 ? 3 : 2) void gram_sym_tiles_split_kernel(const float* __restrict__ Z, float* __restrict__ E, int B, int N, int D, int nt) {
    constexpr int GT = 64, BK = 32, SPLD = BK + 16;
    typedef typename std::conditional<SPL == 2, _Float16, __bf16>::type half_t;
    typedef typename std::conditional<SPL == 2, f16x8, bf16x8>::type frag_t;
    __shared__ __attribute__((aligned(16))) half_t As[2][SPL][GT * SPLD];     // [buffer][plane]
    __shared__ __attribute__((aligned(16))) half_t Bs[2][SPL][GT * SPLD];
    const int ntile = nt * (nt + 1) / 2;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int b = (slot / ntile) * 8 + xcd;
    if (b >= B) return;
    int tm = 0, rem = slot % ntile;
    while (rem > tm) { rem -= tm + 1; ++tm; }            // lower triangle, row by row: tile (tm, tn = rem), tn <= tm
    const int tn = rem;
    const bool diag = tn == tm;
    const float* Zb = Z + (size_t)b * N * D;
    float* Eb = E + (size_t)b * N * N;
    const int m0 = tm * GT, n0 = tn * GT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r16 = lane & 15, q = lane >> 4;
    const int lr = tid >> 3, lc = (tid & 7) * 4;
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Zb), 0, N * D * 4, 0x00020000);
    float4 ra[2], rb[2];
    auto gload = [&](int k0) {
        const bool in = k0 + lc < D;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int rowa = m0 + lr + 32 * h, rowb = n0 + lr + 32 * h;
            const auto va = __builtin_amdgcn_raw_buffer_load_b128(zr, (in && rowa < N) ? (rowa * D + lc) * 4 : 0x7ffffff0, k0 * 4, 0);
            ra[h] = make_float4(__uint_as_float(va[0]), __uint_as_float(va[1]), __uint_as_float(va[2]), __uint_as_float(va[3]));
            if (!diag) {
                const auto vb = __builtin_amdgcn_raw_buffer_load_b128(zr, (in && rowb < N) ? (rowb * D + lc) * 4 : 0x7ffffff0, k0 * 4, 0);
                rb[h] = make_float4(__uint_as_float(vb[0]), __uint_as_float(vb[1]), __uint_as_float(vb[2]), __uint_as_float(vb[3]));
            }
        }
    };
    auto put = [&](half_t (&planes)[SPL][GT * SPLD], const float4& v, int row) {
        if constexpr (SPL == 2) {
            f16x4 hh, mm;
            split2h(v, 32768.f, hh, mm);
            *reinterpret_cast<f16x4*>(&planes[0][row * SPLD + lc]) = hh;
            *reinterpret_cast<f16x4*>(&planes[1][row * SPLD + lc]) = mm;
        } else {
            bf16x4 hh, mm, ll;
            split3(v, hh, mm, ll);
            *reinterpret_cast<bf16x4*>(&planes[0][row * SPLD + lc]) = hh;
            *reinterpret_cast<bf16x4*>(&planes[1][row * SPLD + lc]) = mm;
            *reinterpret_cast<bf16x4*>(&planes[2][row * SPLD + lc]) = ll;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            put(As[buf], ra[h], lr + 32 * h);
            if (!diag) put(Bs[buf], rb[h], lr + 32 * h);
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (D + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        frag_t af[SPL][2], bf[SPL][2];
#pragma unroll
        for (int pl = 0; pl < SPL; ++pl)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                af[pl][f] = *reinterpret_cast<const frag_t*>(&As[buf][pl][(wm * 32 + 16 * f + r16) * SPLD + 8 * q]);
                bf[pl][f] = *reinterpret_cast<const frag_t*>(&(diag ? As[buf][pl] : Bs[buf][pl])[(wn * 32 + 16 * f + r16) * SPLD + 8 * q]);
            }
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
                if constexpr (SPL == 2) {
                    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][fi], bf[1][fj], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1][fi], bf[0][fj], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][fi], bf[0][fj], t, 0, 0, 0);
                } else {                                 // smallest terms first: mm, hl, lh, hm, mh, hh
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][fi], bf[1][fj], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][fi], bf[2][fj], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2][fi], bf[0][fj], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][fi], bf[1][fj], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][fi], bf[0][fj], t, 0, 0, 0);
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][fi], bf[0][fj], t, 0, 0, 0);
                }
                acc[fi][fj] += t;
            }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    constexpr float UNSCALE = SPL == 2 ? 1.f / (32768.f * 32768.f) : 1.f;
#pragma unroll
    for (int fi = 0; fi < 2; ++fi)
#pragma unroll
        for (int fj = 0; fj < 2; ++fj)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gm = m0 + wm * 32 + fi * 16 + 4 * q + reg, gn = n0 + wn * 32 + fj * 16 + r16;
                if (gm >= N || gn >= N) continue;
                if (diag && gn > gm) continue;           // keep the matrix exactly symmetric
                const float v = acc[fi][fj][reg] * UNSCALE;
                Eb[(size_t)gm * N + gn] = v;
                if (gm != gn) Eb[(size_t)gn * N + gm] = v;
            }
}

// ---------------------------------------------------------------------------------------------------------------------------
// EPISODE-RESIDENT symmetric Gram for 128 < N <= 432 with unit rows (round 4; DKT_GRAM_BIG_EP=0 restores the 64 x 64-tile kernel above).
// The tile kernel stages every 64-row block of Z once per 64 x 64 output tile it takes part in -- 8 split-VALU instructions per MFMA, the matrix
// pipe 20 % busy (profiles/r04/v0_gram_big_pmc.txt).  Here ONE workgroup of 8 waves owns an episode: all N rows of a 32-feature slab are split and
// staged ONCE (two f16 planes, double-buffered), and all NT (NT + 1) / 2 lower 16 x 16 tiles (378 at N = 420) stay in the accumulators of the 8 waves
// (a wave: a run of <= 48 consecutive tiles in row-major order, so the A fragment changes only at a row change).  LDS rows have NO padding (64 B per
// row and plane: 2 x 2 x 27 KB); the 16-byte units of a row are XOR-swizzled with G[(row >> 2) & 3], G = {0, 2, 3, 1}, which makes the
// ds_read_b128 fragment reads conflict-free (every 16-lane group of the instruction then covers all 16 bank quads).
constexpr int BEP_MAXT = 48;
__global__ __launch_bounds__(512, 1) void gram_sym_bigep_f16x2_kernel(const float* __restrict__ Z, float* __restrict__ E, int N, int D) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bep_smem[];
    const int b = blockIdx.x;
    const float* Zb = Z + (size_t)b * N * D;
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = (N + 15) >> 4, NP = 16 * NT, PLANE = NP * 64, BUF = 2 * PLANE;
    const int ntile = NT * (NT + 1) / 2, per = (ntile + 7) >> 3;
    const int t0 = w * per, cnt = max(0, min(per, ntile - t0));
    // first tile of the run: row-major lower triangle, t -> (i, j), j <= i
    int is = 0, js = t0;
    while (js > is) { js -= is + 1; ++is; }
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Zb), 0, N * D * 4, 0x00020000);
    constexpr int NPASS = 7;                               // 432 rows x 8 float4 / 512 threads
    // LDS offset of the staged float4 of pass p: row = row0 + 64 p (so (row >> 2) & 3 does not depend on p), 8-byte half (c4 & 1) of unit (c4 >> 1) ^ G
    const int row0 = tid >> 3, c40 = tid & 7;
    const int voff0 = (row0 * D + 4 * c40) * 4, vstep = 64 * D * 4;        // (per-pass offsets are re-derived in gload: 6 VGPRs the operand prefetch needs)
    const int loff0 = row0 * 64 + ((((c40 >> 1) ^ ((0x78 >> (2 * ((row0 >> 2) & 3))) & 3))) << 4) + ((c40 & 1) << 3);
    float4 rg[NPASS];
    auto gload = [&](const int k0) {
        int vo = voff0;
        asm volatile("" : "+v"(vo));                                       // (opaque: the seven predicated offsets are not hoisted out of the slab loop)
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(zr, (k0 + 4 * c40 < D && row0 + 64 * p < N) ? vo : 0x7ffffff0, k0 * 4 + p * vstep, 0);
            rg[p] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto lstore = [&](const int buf) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if (row0 + 64 * p < NP) {
                f16x4 h, m;
                split2h(rg[p], 32768.f, h, m);
                unsigned char* dst = bep_smem + buf * BUF + loff0 + 4096 * p;
                *reinterpret_cast<f16x4*>(dst) = h;
                *reinterpret_cast<f16x4*>(dst + PLANE) = m;
            }
        }
    };
    f32x4 acc[BEP_MAXT];
#pragma unroll
    for (int u = 0; u < BEP_MAXT; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fl = r16 * 64 + ((q ^ ((0x78 >> (2 * ((r16 >> 2) & 3))) & 3)) << 4);      // this lane's fragment offset inside a 16-row block
    const int nk = (D + 31) >> 5;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * 32);
        const unsigned char* base = bep_smem + buf * BUF + fl;
        int i = is, j = js, cn = cnt;
        asm volatile("" : "+s"(i), "+s"(j), "+s"(cn));      // (opaque: the per-tile coordinates / predicates are recomputed per slab on the SALU, not hoisted into 150 SGPRs)
        f16x8 ah = *reinterpret_cast<const f16x8*>(base + i * 1024), am = *reinterpret_cast<const f16x8*>(base + i * 1024 + PLANE);
        // the B fragments of tile u + 1 are read while tile u multiplies (two register sets; the accumulators take the three plane products directly --
        // the registers of the two-level sum pay for the second set)
        f16x8 bh[2], bm[2];
        bh[0] = *reinterpret_cast<const f16x8*>(base + j * 1024);
        bm[0] = *reinterpret_cast<const f16x8*>(base + j * 1024 + PLANE);
#pragma unroll
        for (int u = 0; u < BEP_MAXT; ++u) {
            if (u < cn) {
                const bool rowchg = j + 1 > i;
                const int jn = rowchg ? 0 : j + 1;
                if (u + 1 < BEP_MAXT) {                     // (unconditional inside the block: a tile past the run re-reads a valid tile and is not used)
                    bh[(u + 1) & 1] = *reinterpret_cast<const f16x8*>(base + jn * 1024);
                    bm[(u + 1) & 1] = *reinterpret_cast<const f16x8*>(base + jn * 1024 + PLANE);
                }
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bm[u & 1], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bh[u & 1], acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[u & 1], acc[u], 0, 0, 0);
                j = jn;
                if (rowchg) {
                    ++i;
                    if (u + 1 < cn) {
                        ah = *reinterpret_cast<const f16x8*>(base + i * 1024);
                        am = *reinterpret_cast<const f16x8*>(base + i * 1024 + PLANE);
                    }
                }
            }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    // ---- store: tile (i, j) and its mirror; a diagonal tile from its lower half only (bitwise symmetric) ----
    const __amdgpu_buffer_rsrc_t er = __builtin_amdgcn_make_buffer_rsrc(E + (size_t)b * N * N, 0, (unsigned)((size_t)N * N * 4), 0x00020000);
    {
        int i = is, j = js;
#pragma unroll
        for (int u = 0; u < BEP_MAXT; ++u) {
            if (u < cnt) {
                const f32x4 v = acc[u] * DKT_F16_UNSCALE_BIG;
                const int gn = 16 * j + r16;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int gm = 16 * i + 4 * q + reg;
                    const bool ok = gm < N && gn < N && (i != j || gn <= gm);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[reg]), er, ok ? (gm * N + gn) * 4 : 0x7ffffff0, 0, 0);
                    if (i == j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[reg]), er, (ok && gn < gm) ? (gn * N + gm) * 4 : 0x7ffffff0, 0, 0);
                }
                if (i != j) {                                // mirror: 4 consecutive columns of row gn (rows of tile row i > j: 16 i + 4 q + 3 < N unless ragged)
                    const int gm0 = 16 * i + 4 * q;
                    if (gm0 + 3 < N) {
                        typedef unsigned bep_u4 __attribute__((ext_vector_type(4)));
                        const bep_u4 uv = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                        __builtin_amdgcn_raw_buffer_store_b128(uv, er, (gn < N) ? (gn * N + gm0) * 4 : 0x7ffffff0, 0, 0);
                    } else {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[reg]), er, (gn < N && gm0 + reg < N) ? (gn * N + gm0 + reg) * 4 : 0x7ffffff0, 0, 0);
                    }
                }
                if (++j > i) { ++i; j = 0; }
            }
        }
    }
}

static int g_bwd_rows8 = -1;
static bool gram_bwd_rows8_enabled() {
    if (g_bwd_rows8 < 0) { const char* v = dkt_variant_env("DKT_GRAM_BWD_ROWS8"); g_bwd_rows8 = (v && v[0] == '0') ? 0 : 1; }
    return g_bwd_rows8 != 0;
}

template <int KS>
void launch_rows(const float* W, const float* Z, float* dZ, int B, int N, int D, const float* sc, hipStream_t st) {
    // 128-row blocks: N > 256 always (three or more blocks per episode); 128 < N <= 256 when the slab loop is long enough to pay for the larger prologue (D >= 128:
    // 0.34 vs 0.41 ms at N = 256, 0.21 vs 0.23 at N = 190, D = 512; at D = 64 the 64-row kernel wins, 0.11 vs 0.125 ms) and the batch fills the GPU with them
    if (gram_bwd_rows8_enabled() && (KS >= 10 || (D >= 128 && (long)B * ((N + 127) / 128) >= 256))) {
        const int nrb = (N + 127) / 128;
        const int grid = 8 * ((B + 7) / 8) * nrb;
        hipLaunchKernelGGL((gram_bwd_rows_f16x2_kernel<KS, 8>), dim3(grid), dim3(512), dkt_lds_pad("DKT_PAD_GRAM_BIG_BWD"), st, W, Z, dZ, B, N, D, sc, nrb);
        return;
    }
    const int nrb = (N + 63) / 64;
    const int grid = 8 * ((B + 7) / 8) * nrb;
    hipLaunchKernelGGL((gram_bwd_rows_f16x2_kernel<KS, 4>), dim3(grid), dim3(256), dkt_lds_pad("DKT_PAD_GRAM_BIG_BWD"), st, W, Z, dZ, B, N, D, sc, nrb);
}

}  // namespace

static int g_big_ep = -1;
static bool gram_big_ep_enabled() {
    if (g_big_ep < 0) { const char* v = dkt_variant_env("DKT_GRAM_BIG_EP"); g_big_ep = (v && v[0] == '0') ? 0 : 1; }
    return g_big_ep != 0;
}
void dkt_gram_big_reload_env();                          // dkt_reload_env(); defined below the switches

// Returns true when the kernel was launched (128 < N <= 448, unit rows, symmetric W, D % 4 == 0, 16-byte aligned Z / dZ).
bool dkt_gram_bwd_big_launch(const float* W, const float* Z, float* dZ, int B, int N, int D, const float* sc, unsigned flags, hipStream_t st) {
    if (N <= 128 || N > 448 || (D & 3) || ((uintptr_t)Z & 15) || ((uintptr_t)dZ & 7)) return false;
    if (!(flags & DKT_GRAM_UNIT_ROWS) || !(flags & DKT_GRAM_W_SYMMETRIC) || !dkt_gram_split_enabled()) return false;
    const int ks = (N + 31) / 32;
    if (ks <= 6) launch_rows<6>(W, Z, dZ, B, N, D, sc, st);
    else if (ks <= 8) launch_rows<8>(W, Z, dZ, B, N, D, sc, st);
    else if (ks <= 10) launch_rows<10>(W, Z, dZ, B, N, D, sc, st);
    else if (ks <= 12) launch_rows<12>(W, Z, dZ, B, N, D, sc, st);
    else launch_rows<14>(W, Z, dZ, B, N, D, sc, st);
    return true;
}

void dkt_gram_big_reload_env() { g_big_ep = -1; g_bwd_rows8 = -1; }

// Returns true when the kernel was launched (symmetric linear Gram, N > 128, D % 4 == 0, 16-byte aligned Z).
bool dkt_gram_sym_big_launch(const float* Z, float* E, int B, int N, int D, bool unit, hipStream_t st) {
    if (N <= 128 || (D & 3) || ((uintptr_t)Z & 15) || !dkt_gram_split_enabled()) return false;
    if (unit && N <= 432 && B >= 64 && gram_big_ep_enabled()) {               // one workgroup per episode: fills the GPU from a few hundred episodes on
        const int NT16 = (N + 15) / 16;
        const size_t lds = (size_t)4 * NT16 * 16 * 64;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)gram_sym_bigep_f16x2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 27 * 16 * 64);
            attr_set = true;
        }
        hipLaunchKernelGGL(gram_sym_bigep_f16x2_kernel, dim3(B), dim3(512), lds, st, Z, E, N, D);
        return true;
    }
    const int nt = (N + 63) / 64;
    const long grid = 8L * ((B + 7) / 8) * (nt * (nt + 1) / 2);
    if (grid > 0x7fffffffL) return false;
    if (unit) hipLaunchKernelGGL(gram_sym_tiles_split_kernel<2>, dim3((unsigned)grid), dim3(256), 0, st, Z, E, B, N, D, nt);
    else hipLaunchKernelGGL(gram_sym_tiles_split_kernel<3>, dim3((unsigned)grid), dim3(256), 0, st, Z, E, B, N, D, nt);
    return true;
}
