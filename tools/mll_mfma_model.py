"""Lane-level numpy model of the wave-per-matrix MFMA marginal-likelihood kernel (csrc/dkt_mll_mfma.hip).

Design aid, not product code: it executes the kernel's algorithm with the exact register layouts of gfx950
(v_mfma_f32_16x16x4_f32 operand / accumulator lane maps, DPP row_newbcast, the replicated column layout of the
diagonal-tile sweep) so that the index logic can be checked on a CPU against numpy's Cholesky / inverse before
any GPU time is spent.  Run:  python tools/mll_mfma_model.py
"""
import numpy as np

L = np.arange(64)
G, Cc = L >> 4, L & 15           # row group, position in the row


def mfma(a, b, c):
    """D = C + A B with A[i][k] in lane (k*16+i), B[k][j] in lane (k*16+j), D[4g+r][j] in lane (g*16+j) reg r."""
    A = np.zeros((16, 4)); Bm = np.zeros((4, 16))
    A[Cc, G] = a
    Bm[G, Cc] = b
    D = A @ Bm
    out = c.copy()
    for r in range(4):
        out[r] += D[4 * G + r, Cc]
    return out


def tile_to_acc(T):
    """16x16 matrix -> accumulator layout [4][64]."""
    return np.stack([T[4 * G + r, Cc] for r in range(4)])


def acc_to_tile(x):
    T = np.zeros((16, 16))
    for r in range(4):
        T[4 * G + r, Cc] = x[r]
    return T


def xty(X, Y, C=None):
    """C + X^T Y on accumulator-layout tiles: register r of X is the A operand, register r of Y the B operand."""
    out = np.zeros((4, 64)) if C is None else C
    for r in range(4):
        out = mfma(X[r], Y[r], out)
    return out


def bcast(x, p):
    """DPP row_newbcast:p"""
    return x[(L & ~15) | p]


def to_columns(S):
    """accumulator layout -> replicated column layout a[i][lane] = T[i][lane & 15] (permlane32/16 swaps on the GPU)."""
    T = acc_to_tile(S)
    return np.stack([T[i, Cc] for i in range(16)])


def sweep(S, pn=None):
    """Diagonal tile: S = -(Schur complement) in accumulator layout -> M = R^-T (acc layout), dv (lane c: raw pivot d_c).
    ONE register per row: lanes c > p of row i hold the (negated) Schur part, lanes c <= p the rows of L^-1 being built;
    both obey x[i] += bcast_p(x[i]) * t with t = x[p] / d, except lane p itself, where the multiplier column turns into a column
    of L^-1: t = 1 / d - 1 = (1 - d) / d (no cancellation: the caller scales the matrix so that every pivot is <= 1).
    pn (last tile only): local index of the augmented pivot, which is forced to 1; the padding pivots are 1 by construction."""
    x = to_columns(S)
    dv = np.ones(64)
    for p in range(16):
        d = -bcast(x[p], p)
        eq = Cc == p
        dv = np.where(eq, d, dv)
        if pn is not None and p == pn:
            d = np.ones(64)
        rs = 1.0 / np.sqrt(d)
        rs2 = rs * rs
        t = np.where(eq, (1.0 - d) * rs2, x[p] * rs2)
        x[p] = np.where(eq, rs, x[p] * rs)
        for i in range(p + 1, 16):
            x[i] = x[i] + bcast(x[i], p) * t
    M = np.zeros((4, 64))
    for r in range(4):
        v = np.choose(G, [x[r], x[4 + r], x[8 + r], x[12 + r]])
        M[r] = np.where(Cc <= 4 * G + r, v, 0.0)
    return M, dv, x


def run(N, K, rvec):
    """Full algorithm on the (N+1)-augmented, padded matrix.  Returns logdet, quad, alpha, K^-1 - alpha alpha^T (N x N)."""
    NT = (N + 1 + 15) // 16
    NP = 16 * NT
    pN = N - 16 * (NT - 1)
    # power-of-4 scaling: pivots <= 1 (K / kappa, r / sqrt(kappa): w, the quadratic form and the signs are unchanged)
    e = int(np.frexp(np.diag(K).max())[1])
    m = max(0, (e + 1) // 2)
    kappa = 4.0 ** m
    Sfull = np.zeros((NP, NP))
    Sfull[:N, :N] = -K / kappa
    Sfull[:N, N] = -rvec / 2.0 ** m
    Sfull[N, :N] = -rvec / 2.0 ** m
    for p in range(N + 1, NP):
        Sfull[p, p] = -1.0
    tile = lambda i, j: tile_to_acc(Sfull[16 * i:16 * i + 16, 16 * j:16 * j + 16])
    T = {(i, j): tile(i, j) for i in range(NT) for j in range(i, NT)}
    negI = tile_to_acc(-np.eye(16))
    Md = {}
    logdet = 0.0
    quad = None
    # ---- phase 1: K' = R^T R, tiles hold S = -(Schur) until they become R ----
    for k in range(NT):
        last = k == NT - 1
        M, dv, _ = sweep(T[(k, k)], pN if last else None)
        Md[k] = M
        valid = (16 * k + Cc) < N
        logdet += np.sum(np.log(dv[:16])[valid[:16]])
        if last:
            quad = -dv[pN]                       # the augmented pivot met S[N][N] = +|w|^2
        nV = xty(M, negI)                        # M^T (-I) = -V_kk
        for j in range(k + 1, NT):
            T[(k, j)] = xty(nV, T[(k, j)])       # R_kj = (-V_kk)^T S_kj
        for i in range(k + 1, NT):
            for j in range(i, NT):
                T[(i, j)] = xty(T[(k, i)], T[(k, j)], T[(i, j)])     # S_ij += R_ki^T R_kj
    # ---- phase 2: M = R^-T, M_ji (j > i) overwrites slot (i, j) ----
    for j in range(1, NT):
        nV = xty(Md[j], negI)
        for i in range(j):
            Q = xty(T[(i, j)], Md[i])            # k = i term: R_ij^T M_ii
            for k in range(i + 1, j):
                Q = xty(T[(k, j)], T[(i, k)], Q)  # R_kj^T M_ki   (M_ki sits in slot (i, k))
            T[(i, j)] = xty(nV, Q)               # M_ji = -V_jj^T Q
    # alpha = -(row N of M)
    alpha = np.zeros(NP)
    for i in range(NT - 1):
        t = acc_to_tile(T[(i, NT - 1)])          # M_{NT-1, i}: rows = block NT-1, cols = block i
        alpha[16 * i:16 * i + 16] = -t[pN, :]
    alpha[16 * (NT - 1):] = -acc_to_tile(Md[NT - 1])[pN, :]
    alpha = alpha[:N]
    # ---- phase 3: P''_ij = sum_{k >= j} Mflip_ki^T M_kj, in place, result in slot (i, j), diagonal in Md ----
    sg = np.stack([np.where(4 * G + r == pN, -1.0, 1.0) for r in range(4)])
    Mt = lambda k, i: Md[k] if k == i else T[(i, k)]
    for j in range(NT):
        for i in list(range(j)) + [j]:
            acc = None
            for k in range(j, NT):
                A = Mt(k, i)
                if k == NT - 1:
                    A = A * sg
                acc = xty(A, Mt(k, j), acc)
            if i == j:
                Md[j] = acc
            else:
                T[(i, j)] = acc
    P = np.zeros((NP, NP))
    for j in range(NT):
        P[16 * j:16 * j + 16, 16 * j:16 * j + 16] = acc_to_tile(Md[j])
        for i in range(j):
            P[16 * i:16 * i + 16, 16 * j:16 * j + 16] = acc_to_tile(T[(i, j)])
            P[16 * j:16 * j + 16, 16 * i:16 * i + 16] = acc_to_tile(T[(i, j)]).T
    return logdet + N * np.log(kappa), quad, alpha / 2.0 ** m, P[:N, :N] / kappa


def main():
    rng = np.random.default_rng(0)
    for N in (5, 15, 16, 19, 25, 31, 85, 105, 111, 112, 127):
        Z = rng.standard_normal((N, 40))
        Z /= np.linalg.norm(Z, axis=1, keepdims=True)
        K = (0.7 * Z @ Z.T + 0.1 * np.eye(N)) * (37.0 if N % 2 else 1.0)
        r = rng.standard_normal(N)
        logdet, quad, alpha, P = run(N, K, r)
        Ki = np.linalg.inv(K)
        a_ref = Ki @ r
        err = [abs(logdet - np.linalg.slogdet(K)[1]), abs(quad - r @ a_ref), np.abs(alpha - a_ref).max(),
               np.abs(P - (Ki - np.outer(a_ref, a_ref))).max()]
        print("N=%3d  logdet %.2e  quad %.2e  alpha %.2e  P %.2e" % (N, *err))
        assert max(err) < 1e-9


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------------------------------------------------------------------
# Large-N tiled path (csrc/dkt_mll_tiled.hip): the same three phases LEFT-looking over a tile array in memory
# (slot (i, j), i <= j; diagonal slot (i, i) holds M_ii = R_ii^-T after phase 1), block rows / block columns of RB tiles.
def run_tiled(N, K, rvec, RB=4):
    NT = (N + 1 + 15) // 16
    NP = 16 * NT
    pN = N - 16 * (NT - 1)
    e = int(np.frexp(np.diag(K).max())[1])
    m = max(0, (e + 1) // 2)
    kappa = 4.0 ** m
    Sfull = np.zeros((NP, NP))
    Sfull[:N, :N] = -K / kappa
    Sfull[:N, N] = -rvec / 2.0 ** m
    Sfull[N, :N] = -rvec / 2.0 ** m
    for p in range(N + 1, NP):
        Sfull[p, p] = -1.0
    form = lambda i, j: tile_to_acc(Sfull[16 * i:16 * i + 16, 16 * j:16 * j + 16])
    negI = tile_to_acc(-np.eye(16))
    G_ = {}                                              # the tile array in memory
    logdet, quad = 0.0, None
    # ---- K1: factorisation, block row I ----
    for i0 in range(0, NT, RB):
        rows = range(i0, min(i0 + RB, NT))
        acc = {(i, j): form(i, j) for i in rows for j in range(i0, NT)}        # (the strictly lower part of the diagonal block is unused)
        for kt in range(i0):
            for i in rows:
                for j in range(i0, NT):
                    acc[(i, j)] = xty(G_[(kt, i)], G_[(kt, j)], acc[(i, j)])
        for i in rows:
            last = i == NT - 1
            M, dv, _ = sweep(acc[(i, i)], pN if last else None)
            valid = (16 * i + Cc) < N
            logdet += np.sum(np.log(dv[:16])[valid[:16]])
            if last:
                quad = -dv[pN]
            nV = xty(M, negI)
            for j in range(i + 1, NT):
                acc[(i, j)] = xty(nV, acc[(i, j)])
            acc[(i, i)] = M
            for i2 in rows:
                if i2 > i:
                    for j in range(i2, NT):
                        acc[(i2, j)] = xty(acc[(i, i2)], acc[(i, j)], acc[(i2, j)])
        for i in rows:
            for j in range(i, NT):
                G_[(i, j)] = acc[(i, j)]
    # ---- K2: M = R^-T, block column J; slot (i, j) <- M_ji ----
    alpha = np.zeros(NP)
    for j0 in range(0, NT, RB):
        cols = range(j0, min(j0 + RB, NT))
        acc = {(i, j): np.zeros((4, 64)) for j in cols for i in range(j)}
        Rblk = {(k, j): G_[(k, j)] for j in cols for k in range(j0, j)}         # diagonal-block R tiles, read before anything is stored
        for k in range(j0):                                                     # previous columns are final
            for j in cols:
                for i in range(k + 1):
                    acc[(i, j)] = xty(G_[(k, j)], G_[(i, k)], acc[(i, j)])     # slot (i, i) = M_ii, slot (i, k) = M_ki
        fin = {}
        for j in cols:
            for k in range(j0, j):
                for i in range(k + 1):
                    Mki = G_[(k, k)] if i == k else fin[(i, k)]
                    acc[(i, j)] = xty(Rblk[(k, j)], Mki, acc[(i, j)])
            nV = xty(G_[(j, j)], negI)
            for i in range(j):
                fin[(i, j)] = xty(nV, acc[(i, j)])
        for (i, j), v in fin.items():
            G_[(i, j)] = v
    for i in range(NT - 1):
        alpha[16 * i:16 * i + 16] = -acc_to_tile(G_[(i, NT - 1)])[pN, :]
    alpha[16 * (NT - 1):] = -acc_to_tile(G_[(NT - 1, NT - 1)])[pN, :]
    # ---- K3: P''_ij = sum_{k >= j} flip(M_ki)^T M_kj, block column J of the OUTPUT (a separate array) ----
    sg = np.stack([np.where(4 * G + r == pN, -1.0, 1.0) for r in range(4)])
    P = np.zeros((NP, NP))
    for j0 in range(0, NT, RB):
        for j in range(j0, min(j0 + RB, NT)):
            for i in range(j + 1):
                acc = np.zeros((4, 64))
                for k in range(j, NT):
                    A = G_[(i, k)] * (sg if k == NT - 1 else 1.0)
                    acc = xty(A, G_[(j, k)], acc)
                t = acc_to_tile(acc)
                P[16 * i:16 * i + 16, 16 * j:16 * j + 16] = t
                P[16 * j:16 * j + 16, 16 * i:16 * i + 16] = t.T
    return logdet + N * np.log(kappa), quad, alpha[:N] / 2.0 ** m, P[:N, :N] / kappa


def main_tiled():
    rng = np.random.default_rng(1)
    for N in (40, 100, 130, 191):
        Z = rng.standard_normal((N, 60))
        Z /= np.linalg.norm(Z, axis=1, keepdims=True)
        K = (0.7 * Z @ Z.T + 0.1 * np.eye(N)) * (5.0 if N % 2 else 1.0)
        r = rng.standard_normal(N)
        logdet, quad, alpha, P = run_tiled(N, K, r)
        Ki = np.linalg.inv(K)
        a_ref = Ki @ r
        err = [abs(logdet - np.linalg.slogdet(K)[1]), abs(quad - r @ a_ref), np.abs(alpha - a_ref).max(),
               np.abs(P - (Ki - np.outer(a_ref, a_ref))).max()]
        print("tiled N=%3d  logdet %.2e  quad %.2e  alpha %.2e  P %.2e" % (N, *err))
        assert max(err) < 1e-8


if __name__ == "__main__" and "tiled" in __import__("sys").argv:
    main_tiled()


# ---------------------------------------------------------------------------------------------------------------------------
# Scaled 2-way f16 split variant (csrc/dkt_mll_h2.hip): the same three phases with every tile product on
# v_mfma_f32_16x16x16_f16.  A tile in the accumulator layout is, with its four registers converted to f16 and packed, a legal A
# operand (lane (g, c), element e  <->  A[c][4g + e] = X[4g + e][c], i.e. A = X^T) AND a legal B operand (B[4g + e][c] = Y[4g + e][c])
# of that instruction, so D += X^T Y is ONE instruction per pair of f16 planes; a stored tile is the pair (h, m) = (f16(x s),
# f16(x s - h)) -- 4 VGPRs, like the fp32 tile -- and a product is hm + mh + hh (the m m term, 2^-22 relative, is dropped).
# Every scale is a power of two and rigorous: Schur complements and R are bounded by 1 after the kappa scaling (s = 2^15),
# |M| = |R^-T| <= mu = sqrt(kappa / (noise + jitter)) (s = 2^15 / mu2), the augmented column is scaled by rho <= 1 / (|r_s| mu2).
# f16 is a FLOATING format: an element keeps 22 significand bits as long as it is within 2^-18 of the bound, so a loose bound
# costs nothing.  float32 arithmetic everywhere the GPU uses it (sweeps, accumulators); `trunc` models an MFMA whose final
# rounding truncates.
F32 = np.float32


def _f16_in(v):
    """an MFMA input: f16 subnormals are flushed"""
    v = np.asarray(v, dtype=np.float16)
    return np.where(np.abs(v.astype(F32)) < 2.0 ** -14, np.float16(0), v)


def mfma16(a4, b4, c, trunc=False):
    """v_mfma_f32_16x16x16_f16: a4 / b4 = [4][64] f16 (lane (g, c), element e: A[c][4g + e], B[4g + e][c]); c = [4][64] f32."""
    A = np.zeros((16, 16)); Bm = np.zeros((16, 16))
    a4 = _f16_in(a4).astype(np.float64); b4 = _f16_in(b4).astype(np.float64)
    for e in range(4):
        A[Cc, 4 * G + e] = a4[e]
        Bm[4 * G + e, Cc] = b4[e]
    D = A @ Bm
    out = np.empty((4, 64), dtype=F32)
    for r in range(4):
        s = c[r].astype(np.float64) + D[4 * G + r, Cc]
        if trunc:
            f = s.astype(F32)
            f = np.where(np.abs(f.astype(np.float64)) > np.abs(s), np.nextafter(f, F32(0)), f)
            out[r] = f
        else:
            out[r] = s.astype(F32)
    return out


def split2(x, scale):
    xs = (np.asarray(x, dtype=F32) * F32(scale)).astype(F32)
    h = xs.astype(np.float16)
    m = (xs - h.astype(F32)).astype(F32).astype(np.float16)
    assert np.all(np.isfinite(h.astype(F32))), "f16 overflow: a scale bound is violated"
    return h, m


def xtyh(X, Y, C=None, trunc=False):
    """C + X^T Y on split tiles, small terms first"""
    out = np.zeros((4, 64), dtype=F32) if C is None else C
    out = mfma16(X[0], Y[1], out, trunc)
    out = mfma16(X[1], Y[0], out, trunc)
    return mfma16(X[0], Y[0], out, trunc)


def sweep32(S, pn=None, om2_min=1.0):
    """the diagonal-tile sweep in float32 (the GPU's arithmetic).  The augmented pivot is forced to omega^2 = the power of 4 above
    max(|raw pivot|, om2_min) -- the raw value is rho^2 |w|^2 -- which lifts row N of M (-rho alpha_s^T / omega, M_NN = 1 / omega) to
    within |M|'s bound: |rho alpha_s| / omega <= mu rho |w| / omega <= mu, 1 / omega <= mu2."""
    x = to_columns(S).astype(F32)
    dv = np.ones(64, dtype=F32)
    one = F32(1.0)
    for p in range(16):
        d = -bcast(x[p], p)
        eq = Cc == p
        dv = np.where(eq, d, dv)
        if pn is not None and p == pn:
            ex = int(np.frexp(max(abs(float(d[0])), om2_min))[1])
            om2 = 4.0 ** ((ex + 1) >> 1)
            d = np.full(64, om2, dtype=F32)
        rs = (one / np.sqrt(d)).astype(F32)
        rs2 = rs * rs
        t = np.where(eq, (one - d) * rs2, x[p] * rs2).astype(F32)
        x[p] = np.where(eq, rs, x[p] * rs).astype(F32)
        for i in range(p + 1, 16):
            x[i] = (x[i] + bcast(x[i], p) * t).astype(F32)
    M = np.zeros((4, 64), dtype=F32)
    for r in range(4):
        v = np.choose(G, [x[r], x[4 + r], x[8 + r], x[12 + r]])
        M[r] = np.where(Cc <= 4 * G + r, v, 0.0)
    return M, dv, (np.sqrt(om2) if pn is not None else 1.0)


def run_h2(N, K, rvec, lam_lb, trunc=False):
    """K (float64, symmetric positive definite with lambda_min >= lam_lb = noise + jitter), r -> logdet, quad, alpha, K^-1 - alpha alpha^T."""
    NT = (N + 1 + 15) // 16
    NP = 16 * NT
    pN = N - 16 * (NT - 1)
    K = K.astype(F32).astype(np.float64)
    e = int(np.frexp(np.diag(K).max())[1])
    msc = max(0, (e + 1) // 2)
    kappa = 4.0 ** msc
    e_mu = (int(np.frexp(kappa / lam_lb)[1]) + 1) >> 1                 # mu2 = 2^e_mu >= sqrt(kappa / lam_lb)
    sM = 2.0 ** (15 - e_mu)
    r2 = float(np.sum(rvec ** 2)) / kappa * 4.0 ** e_mu
    e_r = (int(np.frexp(r2)[1]) + 1) >> 1 if r2 > 0 else 0              # rho = 2^-e_r <= 1 / (|r_s| mu2)
    rho = 2.0 ** -e_r
    T30, S15 = 2.0 ** 30, 2.0 ** 15
    Sfull = np.zeros((NP, NP))
    Sfull[:N, :N] = -K / kappa
    Sfull[:N, N] = -rvec / 2.0 ** msc * rho
    Sfull[N, :N] = -rvec / 2.0 ** msc * rho
    for p in range(N + 1, NP):
        Sfull[p, p] = -1.0
    Sfull = (Sfull * T30).astype(F32)                                    # tiles hold 2^30 S
    T = {(i, j): tile_to_acc(Sfull[16 * i:16 * i + 16, 16 * j:16 * j + 16]).astype(F32) for i in range(NT) for j in range(i, NT)}
    negI = np.stack([np.where(4 * G + r == Cc, -1.0, 0.0) for r in range(4)]).astype(np.float16)
    zero16 = np.zeros((4, 64), dtype=np.float16)
    Md = {}
    logdet, quad, alast = 0.0, None, None

    def transpose_neg(Msp):           # -M^T, plane by plane, through the matrix pipe (exact: one non-zero term per element)
        return tuple(mfma16(Msp[pl], negI, np.zeros((4, 64), dtype=F32)).astype(np.float16) for pl in range(2))

    # ---- phase 1 ----
    for k in range(NT):
        last = k == NT - 1
        M, dv, om = sweep32((T[(k, k)] * F32(2.0 ** -30)).astype(F32), pN if last else None, 4.0 ** -e_mu)
        valid = (16 * k + Cc) < N
        logdet += np.sum(np.log(dv[:16].astype(np.float64))[valid[:16]])
        if last:
            quad = -float(dv[pN]) / rho ** 2
            alast = -acc_to_tile(M)[pN, :] * om                           # alpha_s rho, last segment (fp32, before the split)
        Md[k] = split2(M, sM)
        nV = transpose_neg(Md[k])
        for j in range(k + 1, NT):
            acc = xtyh(nV, split2(T[(k, j)], 2.0 ** -15), None, trunc)     # sM 2^15 R_kj
            T[(k, j)] = split2(acc, 1.0 / sM)                              # R_kj 2^15, stored split
        for i in range(k + 1, NT):
            for j in range(i, NT):
                T[(i, j)] = xtyh(T[(k, i)], T[(k, j)], T[(i, j)], trunc)   # 2^30 S_ij += (2^15 R_ki)^T (2^15 R_kj)
    # ---- phase 2 ----
    alpha = np.zeros(NP)
    for j in range(1, NT):
        nV = transpose_neg(Md[j])
        for i in range(j):
            Q = xtyh(T[(i, j)], Md[i], None, trunc)
            for k in range(i + 1, j):
                Q = xtyh(T[(k, j)], T[(i, k)], Q, trunc)                   # 2^15 sM Q
            acc = xtyh(nV, split2(Q, 2.0 ** -19), None, trunc)             # sM (sM / 16) M_ji
            if j == NT - 1:
                alpha[16 * i:16 * i + 16] = -acc_to_tile(acc)[pN, :] * (16.0 / sM ** 2) * om
            T[(i, j)] = split2(acc, 16.0 / sM)                             # sM M_ji, stored split
    alpha[16 * (NT - 1):] = alast
    alpha_s = alpha[:N] / rho                                              # alpha of the kappa-scaled system
    # ---- phase 3: row N of M (the augmented row) is zeroed, P = M^T M is K_s^-1; the rank-one term is added in fp32 ----
    rowmask = np.stack([np.where(4 * G + r == pN, 0.0, 1.0) for r in range(4)]).astype(np.float16)
    for i in range(NT - 1):
        T[(i, NT - 1)] = tuple(pl * rowmask for pl in T[(i, NT - 1)])
    Md[NT - 1] = tuple(pl * rowmask for pl in Md[NT - 1])
    Mt = lambda k, i: Md[k] if k == i else T[(i, k)]
    for j in range(NT):
        for i in list(range(j)) + [j]:
            acc = None
            for k in range(j, NT):
                acc = xtyh(Mt(k, i), Mt(k, j), acc, trunc)
            if i == j:
                Md[j] = acc
            else:
                T[(i, j)] = acc
    P = np.zeros((NP, NP))
    for j in range(NT):
        P[16 * j:16 * j + 16, 16 * j:16 * j + 16] = acc_to_tile(Md[j])
        for i in range(j):
            P[16 * i:16 * i + 16, 16 * j:16 * j + 16] = acc_to_tile(T[(i, j)])
            P[16 * j:16 * j + 16, 16 * i:16 * i + 16] = acc_to_tile(T[(i, j)]).T
    Pk = P[:N, :N] / sM ** 2 - np.outer(alpha_s, alpha_s)
    return logdet + N * np.log(kappa), quad, alpha_s / 2.0 ** msc, Pk / kappa


def main_h2():
    rng = np.random.default_rng(2)
    cases = [(105, 1600, 0.69, 0.1, 1.0), (105, 64, 0.69, 0.1, 1.0), (85, 512, 0.69, 0.1, 1.0), (25, 64, 0.69, 0.1, 1.0), (19, 2916, 0.69, 0.69, 30.0),
             (105, 64, 0.69, 1e-4, 1.0), (105, 1600, 30.0, 1e-2, 1.0), (60, 40, 0.7, 1e-6, 1.0), (31, 10, 5.0, 0.1, 100.0), (127, 200, 1.0, 0.05, 1.0)]
    for trunc in (False, True):
        for (N, D, s, nz, ysc) in cases:
            Z = rng.standard_normal((N, D)); Z /= np.linalg.norm(Z, axis=1, keepdims=True)
            K = s * (Z @ Z.T) + nz * np.eye(N)
            K = K.astype(F32).astype(np.float64)
            r = np.where(rng.random(N) < 0.2, 1.0, -1.0) * ysc - 0.03
            logdet, quad, alpha, P = run_h2(N, K, r, nz, trunc)
            Ki = np.linalg.inv(K); a_ref = Ki @ r
            logp = -0.5 * quad - 0.5 * logdet; logp_ref = -0.5 * r @ a_ref - 0.5 * np.linalg.slogdet(K)[1]
            Pref = Ki - np.outer(a_ref, a_ref)
            print("h2%s N=%3d D=%4d s=%.2f nz=%.0e y=%g: logp rel %.1e  logdet %.1e  quad rel %.1e  alpha rel %.1e  P rel-F %.1e" % (
                " trunc" if trunc else "", N, D, s, nz, ysc, abs(logp - logp_ref) / abs(logp_ref), abs(logdet - np.linalg.slogdet(K)[1]),
                abs(quad - r @ a_ref) / abs(r @ a_ref), np.abs(alpha - a_ref).max() / np.abs(a_ref).max(), np.linalg.norm(P - Pref) / np.linalg.norm(Pref)))


if __name__ == "__main__" and "h2" in __import__("sys").argv:
    main_h2()
