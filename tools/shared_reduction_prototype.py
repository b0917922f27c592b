"""Design study: stop factorising the same matrix C times.

The C class matrices of a linear-kernel episode are shifts of ONE matrix, K_c = s_c (E + mu_c I), mu_c = noise_c / s_c (methods/DKT.py:148-149 hands every class
model the same z_train; :346-347 freezes the noise).  One orthogonal reduction per episode, E = Q T Q^T with T tridiagonal (Householder) or banded (first stage of a
two-stage reduction), makes every class an O(N) / O(N b^2) factorisation of T + mu_c I:
    log det K_c = N log s_c + sum_i log d_i                      (pivots of the LDL^T / banded Cholesky; failure = non-positive pivot -> jitter on noise_c)
    alpha_c     = Q (T + mu_c)^-1 Q^T r_c / s_c                  (O(N^2))
    W           = 0.5 sum_c w_c s_c alpha_c alpha_c^T - 0.5 Q [sum_c w_c (T + mu_c)^-1] Q^T          (ONE similarity product per episode)
This script is the float32 prototype of both variants against the float64 oracle on the cfg4 / cfg2 shapes (log-likelihood 1e-4, gradients 1e-3, jitter / failure
equivalence) and prints the flop / byte budget DESIGN.md 6.6 discusses.  numpy only; nothing here is product code.

    python tools/shared_reduction_prototype.py [--n 420 --d 512 --c 20 --band 32]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dkt_oracle as O  # noqa: E402  (tools/ may use the oracle: it is the checker here)


def band_reduce(a, b):
    """Symmetric A (float32) -> (B, Q) with A = Q B Q^T, B of half-bandwidth b (b = 1: tridiagonal): unblocked Householder, one reflector per column."""
    a = a.astype(np.float32).copy()
    n = a.shape[0]
    q = np.eye(n, dtype=np.float32)
    for k in range(n - b - 1):
        x = a[k + b:, k].copy()
        nx = np.float32(np.linalg.norm(x))
        if nx == 0.0:
            continue
        v = x.copy()
        v[0] += np.float32(np.copysign(nx, x[0]))
        v /= np.float32(np.linalg.norm(v))
        # A <- H A H, H = I - 2 v v^T acting on rows / columns k + b ..
        a[k + b:, :] -= np.float32(2.0) * np.outer(v, v @ a[k + b:, :])
        a[:, k + b:] -= np.float32(2.0) * np.outer(a[:, k + b:] @ v, v)
        q[:, k + b:] -= np.float32(2.0) * np.outer(q[:, k + b:] @ v, v)
    # clean what rounding left outside the band
    i, j = np.indices(a.shape)
    a[np.abs(i - j) > b] = 0.0
    return 0.5 * (a + a.T), q


def band_cholesky(t, b):
    """Lower banded Cholesky in float32; returns (L, fail_at) with fail_at = 1-based index of the first non-positive pivot (0 = ok)."""
    n = t.shape[0]
    l = np.zeros_like(t)
    for j in range(n):
        lo = max(0, j - b)
        d = t[j, j] - np.dot(l[j, lo:j], l[j, lo:j])
        if not d > 0:
            return l, j + 1
        l[j, j] = np.sqrt(d)
        hi = min(n, j + b + 1)
        for i in range(j + 1, hi):
            lo2 = max(0, i - b)
            l[i, j] = (t[i, j] - np.dot(l[i, lo2:j], l[j, lo2:j])) / l[j, j]
    return l, 0


def episode(z, y, sv, mean, noise, cw, b, jitter0=1e-6, max_tries=3):
    """float32 prototype of one training episode through ONE reduction; returns logp[C], alpha[C,N], W[N,N], jitter[C], info[C]."""
    import scipy.linalg as sla
    z = z.astype(np.float32)
    e = z @ z.T
    t, q = band_reduce(e, b)
    n, c = e.shape[0], len(sv)
    logp = np.zeros(c, np.float32)
    alpha = np.zeros((c, n), np.float32)
    jit_used = np.zeros(c, np.float32)
    info = np.zeros(c, np.int32)
    ssum = np.zeros((n, n), np.float32)
    w = np.zeros((n, n), np.float32)
    eye = np.eye(n, dtype=np.float32)
    for k in range(c):
        for attempt in range(max_tries + 1):
            jit = np.float32(0.0 if attempt == 0 else jitter0 * 10.0 ** (attempt - 1))
            mu = np.float32((noise[k] + jit) / sv[k])
            l, fail = band_cholesky(t + mu * eye, b)
            if fail == 0:
                break
        jit_used[k], info[k] = jit, fail
        if fail:
            logp[k] = np.nan
            continue
        r = (y[k] - mean[k]).astype(np.float32)
        u = q.T @ r
        x = sla.solve_triangular(l, u, lower=True).astype(np.float32)
        quad = np.float32(x @ x) / np.float32(sv[k])
        logdet = np.float32(n * np.log(sv[k]) + 2.0 * np.log(np.diag(l)).sum())
        logp[k] = -0.5 * quad - 0.5 * logdet - 0.5 * n * np.log(2.0 * np.pi)
        tinv_u = sla.solve_triangular(l.T, x, lower=False).astype(np.float32)
        alpha[k] = (q @ tinv_u) / np.float32(sv[k])
        linv = sla.solve_triangular(l, eye, lower=True).astype(np.float32)       # (T + mu)^-1 = L^-T L^-1: dense inverse of a banded SPD matrix
        ssum += np.float32(cw[k]) * (linv.T @ linv)
        w += np.float32(0.5 * cw[k] * sv[k]) * np.outer(alpha[k], alpha[k])
    w -= 0.5 * (q @ ssum @ q.T)
    return logp, alpha, 0.5 * (w + w.T), jit_used, info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=420)
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--c", type=int, default=20)
    ap.add_argument("--band", type=int, nargs="*", default=[1, 32])
    ap.add_argument("--corr", type=int, default=0)
    args = ap.parse_args()
    n, d, c = args.n, args.d, args.c
    z = O.synthetic_features(1, n, d, 5, args.corr)[0]
    hyp = O.perturbed_hypers(c, 9)
    y = O.one_vs_rest_targets(c, n // c)
    cw = np.full(c, -1.0 / (c * n))
    e64 = z.astype(np.float32).astype(np.float64) @ z.astype(np.float32).astype(np.float64).T
    res = O.mll_terms(e64, y, hyp.outputscale, hyp.mean, hyp.noise)
    w_ref, _, _, _ = O.mll_grads(e64, res, hyp.outputscale, hyp.noise, cw)
    print("shape N=%d D=%d C=%d corr=%d  cond(K_0)=%.0f" % (n, d, c, args.corr, np.linalg.cond(hyp.outputscale[0] * e64 + hyp.noise[0] * np.eye(n))))
    for b in args.band:
        logp, alpha, w, jit, info = episode(z, y, hyp.outputscale, hyp.mean, hyp.noise, cw, b)
        rel = lambda a, r: float(np.linalg.norm(a - r) / np.linalg.norm(r))        # noqa: E731
        print("  band %2d: logp rel err %.2e (tol 1e-4)  alpha rel-L2 %.2e  W rel-L2 %.2e (tol 1e-3)  dZ rel-L2 %.2e  info %s  jitter %s"
              % (b, np.abs((logp - res.logp) / res.logp).max(), rel(alpha, res.alpha), rel(w, w_ref),
                 rel((w + w.T) @ z, (w_ref + w_ref.T) @ z), "ok" if not info.any() else info.tolist(), "0" if not jit.any() else jit.tolist()))
    # failure / jitter equivalence: a noise-free, rank-deficient episode (N > D) must fail at attempt 0 and succeed with jitter, as the N x N Cholesky does
    zs = O.synthetic_features(1, 60, 16, 3, 0)[0]
    ys = O.one_vs_rest_targets(3, 20)
    for b in args.band:
        lp, _, _, jit, info = episode(zs, ys, np.ones(3), np.zeros(3), np.array([0.0, 0.1, 0.0]), np.full(3, -1.0 / 180), min(b, 8))
        print("  band %2d, rank-deficient E with noise (0, 0.1, 0): jitter %s info %s (N x N reference: psd_safe_cholesky retries classes 0 and 2)" % (min(b, 8), jit.tolist(), info.tolist()))
    n3 = float(n) ** 3
    print("flops per episode (N^3 = %.1f MF):" % (n3 / 1e6))
    print("  today, C factorise + invert + M^T M : C * 4/3 N^3           = %6.1f N^3" % (c * 4.0 / 3.0))
    for b in (1, 32):
        inv = 2.0 * b * c / n if b > 1 else 2.0 * c / n          # dense inverses of the C banded matrices: ~ 2 b N^2 each (tridiagonal: two O(N^2) sequences)
        tot = 4.0 / 3.0 + 4.0 / 3.0 + 3.0 + inv
        print("  one reduction, band %2d: reduce 4/3 + form Q 4/3 + C inverses %.2f + Q S Q^T 3 = %5.2f N^3  (%.1f x fewer)" % (b, inv, tot, c * 4.0 / 3.0 / tot))


if __name__ == "__main__":
    main()
