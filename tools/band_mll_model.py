"""Block-level numpy model of csrc/dkt_mll_band.hip: the marginal likelihood of the C class models of a shared-E episode through ONE
orthogonal reduction of E to block-tridiagonal form (16 x 16 blocks), B = Q^T E Q, instead of C dense factorisations.

    K_c = sv_c (E + mu_c I),  mu_c = noise_c / sv_c            (methods/DKT.py:148-149: every class model gets the same z_train; :346-347 noise frozen)
    log det K_c = N log sv_c + sum log pivots of the block LDL^T of B + mu_c I
    alpha_c     = Q a_c / sv_c,   a_c = (B + mu_c)^-1 Q^T r_c
    W           = Q [ sum_c 0.5 cw_c (a_c a_c^T / sv_c - (B + mu_c)^-1) ] Q^T

This file states the index logic of the kernels step by step (the same loops, the same operand orientations: every product is X^T Y on 16 x 16 blocks),
in float64 or float32; tests/test_band_model.py holds it to the oracle.  Stages:
    reduce   : panel k = block column k, rows >= 16 (k + 1): Householder QR -> V_k (unit lower trapezoidal), tau; T_k by the larft recurrence from V^T V;
               A <- H_k^T A H_k as  X = A V, S = V^T X, Y = X Th^T - 0.5 V (Th S Th^T) with Th = T^T, A -= V Y^T + Y V^T;  U <- H_k^T U (U = [r_c] columns)
    classes  : per class the block LDL^T chain  P_j = D_j + mu - G_{j-1} S_{j-1}^T,  G_j = S_j P_j^-1,  forward / backward substitution for a_c,
               Z_jj = P_j^-1 + G_j^T Z_{j+1,j+1} G_j,   then the column chains  Z_{j,i} = -G_j^T Z_{j+1,i}  (j < i) accumulated over the classes
    back     : M <- H_k M H_k^T for k = last .. 0 (the same two-sided kernel with Th = T), A <- H_k A (columns a_c)
"""
import numpy as np

TS = 16


def householder_panel(p):
    """In-place Householder QR of the m x 16 panel p (LAPACK geqr2 conventions: H_j = I - tau_j v_j v_j^T, v_j[j] = 1, R on and above the diagonal).
    Returns (V [m,16] unit lower trapezoidal, tau [16], R [16,16] upper)."""
    m, w = p.shape
    p = p.copy()
    tau = np.zeros(w, p.dtype)
    for j in range(w):
        alpha = p[j, j]
        ss = (p[j + 1:, j] ** 2).sum(dtype=p.dtype)
        if ss == 0:
            tau[j] = 0
            p[j + 1:, j] = 0
            continue
        norm = np.sqrt(alpha * alpha + ss)
        beta = -np.copysign(norm, alpha)
        tau[j] = (beta - alpha) / beta
        scale = 1 / (alpha - beta)
        v = p[j:, j].copy()
        v[0] = 1
        v[1:] *= scale
        w_ = v @ p[j:, j + 1:]                     # v^T P[:, c'] for the later columns
        p[j:, j + 1:] -= tau[j] * np.outer(v, w_)
        p[j, j] = beta
        p[j + 1:, j] = v[1:]
    v = np.tril(p, -1)[:, :w]
    v[np.arange(w), np.arange(w)] = 1
    return v, tau, np.triu(p[:w, :w])


def larft_from_gram(g, tau):
    """T (upper) of H_0 H_1 ... = I - V T V^T from G = V^T V: row i is independent of the other rows.  T[i,i] = tau_i, T[i,j] = -tau_j sum_{l=i}^{j-1} T[i,l] G[l,j]."""
    w = len(tau)
    t = np.zeros((w, w), g.dtype)
    for i in range(w):
        t[i, i] = tau[i]
        for j in range(i + 1, w):
            t[i, j] = -tau[j] * (t[i, i:j] @ g[i:j, j])
    return t


def two_sided(a, v, th, r0, rows_from):
    """a <- (I - V Th V^T) a (I - V Th^T V^T) for symmetric a; V occupies rows r0.. (16 columns); X / Y are built for the rows >= rows_from (the rows and
    columns below it are not touched: forward pass rows_from = r0; backward pass rows_from = 0).  Products as the kernel forms them."""
    x = a[rows_from:, r0:] @ v                      # X = A V                (kernel: Xt_i = sum_j V_j^T A_ji)
    s = v.T @ x[r0 - rows_from:]                    # S = V^T A V
    wm = th @ s @ th.T
    y = x @ th.T                                    # Y = X Th^T - 0.5 V Wm
    y[r0 - rows_from:] -= 0.5 * v @ wm
    a[rows_from:, r0:] -= y @ v.T                   # A -= Y V^T + V Y^T
    a[r0:, rows_from:] -= v @ y.T


def reduce_to_band(e, u):
    """E [NP,NP] (zero padded), U [NP,C] -> (B, V-list, T-list, Q^T U)."""
    a = e.copy()
    u = u.copy()
    npad = a.shape[0]
    nt = npad // TS
    vs, ts = [], []
    for k in range(nt - 2):
        r0 = TS * (k + 1)
        v, tau, r = householder_panel(a[r0:, TS * k:TS * (k + 1)])
        t = larft_from_gram(v.T @ v, tau)
        a[r0:, TS * k:TS * (k + 1)] = 0
        a[r0:r0 + TS, TS * k:TS * (k + 1)] = r
        a[TS * k:TS * (k + 1), r0:] = a[r0:, TS * k:TS * (k + 1)].T
        two_sided(a, v, t.T, r0, r0)                # H^T A H: Th = T^T
        u[r0:] -= v @ (t.T @ (v.T @ u[r0:]))        # U <- H^T U
        vs.append(v)
        ts.append(t)
    return a, vs, ts, u


def class_chain(b, mu, u, n):
    """Block LDL^T of B + mu I (pad rows: unit diagonal).  Returns (logdet, a, G list, Zdiag list, fail_at).  fail_at: 1-based index of the first non-positive pivot."""
    npad = b.shape[0]
    nt = npad // TS
    dt = b.dtype
    g_prev = None
    gs, pinvs, ys = [], [], []
    logdet = dt.type(0)
    fail_at = 0
    for j in range(nt):
        sl = slice(TS * j, TS * (j + 1))
        p = b[sl, sl].copy()
        for r in range(TS):
            p[r, r] = p[r, r] + mu if TS * j + r < n else 1.0
        if j > 0:
            p -= g_prev @ b[TS * (j - 1):TS * j, sl]       # G_{j-1} S_{j-1}^T, S_{j-1}^T = block (j-1, j)
        try:
            l = np.linalg.cholesky(p.astype(np.float64)).astype(dt)
        except np.linalg.LinAlgError:
            return None, None, None, None, TS * j + 1
        piv = np.diag(l) ** 2
        logdet += np.log(piv[:max(0, min(TS, n - TS * j))]).sum(dtype=dt)
        linv = np.linalg.inv(l)
        pinv = linv.T @ linv
        y = u[sl].copy()
        if j > 0:
            y -= g_prev @ ys[-1]
        ys.append(y)
        pinvs.append(pinv)
        if j + 1 < nt:
            s_j = b[TS * (j + 1):TS * (j + 2), sl]
            g_prev = s_j @ pinv
            gs.append(g_prev)
    a = np.zeros(npad, dt)
    zd = [None] * nt
    for j in range(nt - 1, -1, -1):
        sl = slice(TS * j, TS * (j + 1))
        z = pinvs[j] @ ys[j]
        if j + 1 < nt:
            z -= gs[j].T @ a[TS * (j + 1):TS * (j + 2)]
            zd[j] = pinvs[j] + gs[j].T @ zd[j + 1] @ gs[j]
        else:
            zd[j] = pinvs[j]
        a[sl] = z
    return logdet, a, gs, zd, fail_at


def episode(e, y, sv, mean, noise, cw, dtype=np.float64):
    """The whole pipeline for one episode.  e [N,N], y [C,N].  Returns dict(logp, alpha, w, dsv, dmean, dnoise, info)."""
    n = e.shape[0]
    c_ = y.shape[0]
    nt = (n + TS - 1) // TS
    npad = nt * TS
    dt = np.dtype(dtype)
    ep = np.zeros((npad, npad), dt)
    ep[:n, :n] = e
    u = np.zeros((npad, c_), dt)
    u[:n] = (y - mean[:, None]).T
    b, vs, ts, u = reduce_to_band(ep, u)
    m = np.zeros((npad, npad), dt)
    amat = np.zeros((npad, c_), dt)
    logp = np.zeros(c_)
    dsv = np.zeros(c_)
    dnoise = np.zeros(c_)
    info = np.zeros(c_, np.int32)
    for c in range(c_):
        mu = dt.type(noise[c] / sv[c])
        logdet, a, gs, zd, fail = class_chain(b, mu, u[:, c], n)
        if fail:
            info[c] = fail
            logp[c] = np.nan
            continue
        quad = float(u[:n, c] @ a[:n]) / sv[c]
        logp[c] = -0.5 * quad - 0.5 * (n * np.log(sv[c]) + float(logdet)) - 0.5 * n * np.log(2 * np.pi)
        trz = 0.0
        wz = -0.5 * cw[c]
        for i in range(nt):                          # column chain of block column i: Z_ii, then upwards
            z = zd[i]
            si = slice(TS * i, TS * (i + 1))
            m[si, si] += wz * z
            trz += np.trace(z[:max(0, min(TS, n - TS * i)), :max(0, min(TS, n - TS * i))])
            for j in range(i - 1, -1, -1):
                z = -gs[j].T @ z
                m[TS * j:TS * (j + 1), si] += wz * z
        aa = float(a[:n] @ a[:n])
        amat[:, c] = a
        # tr(M_c E), tr(M_c) with M_c = 0.5 (alpha alpha^T - K_c^-1):  alpha^T E alpha = (a.u - mu a.a) / sv^2,  tr(K^-1 E) = (N - mu tr Z) / sv
        dsv[c] = 0.5 * ((float(a[:n] @ u[:n, c]) - mu * aa) / sv[c] ** 2 - (n - mu * trz) / sv[c])
        dnoise[c] = 0.5 * (aa / sv[c] ** 2 - trz / sv[c])
    # mirror the upper block triangle, add the rank-C term, transform back
    for i in range(nt):
        for j in range(i):
            m[TS * i:TS * (i + 1), TS * j:TS * (j + 1)] = m[TS * j:TS * (j + 1), TS * i:TS * (i + 1)].T
    ok = info == 0
    m += (amat[:, ok] * (0.5 * cw[ok] / sv[ok])) @ amat[:, ok].T
    for k in range(len(vs) - 1, -1, -1):
        r0 = TS * (k + 1)
        two_sided(m, vs[k], ts[k], r0, 0)            # H M H^T: Th = T
        amat[r0:] -= vs[k] @ (ts[k] @ (vs[k].T @ amat[r0:]))
    alpha = (amat[:n] / sv).T
    return dict(logp=logp, alpha=alpha, w=m[:n, :n], dsv=dsv, dmean=alpha.sum(1), dnoise=dnoise, info=info, band=b)
