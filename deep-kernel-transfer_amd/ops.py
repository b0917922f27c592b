"""Torch-facing wrappers over the C ABI (include/dkt_abi.h): raw device pointers + the current HIP
stream go down, nothing else.  PyTorch only owns the memory and the stream.

Every function REQUIRES float32 CUDA (ROCm) tensors and raises otherwise -- there is no CPU path.

Autograd surface
  episode_loss_linear(z, y, sv, mean, noise, cls_weight) -> obj[B]   (fused: gram -> mll -> gram_bwd)
  base_matrix(z, kind, lengthscale) -> E[B,N,N]                       (differentiable Gram / RBF)
  mll_objective(e, y, sv, mean, noise, cls_weight) -> obj[B], aux     (differentiable in e, sv, mean, noise)
which replace `-self.mll(self.model(*inputs), targets)` + `.backward()` of the reference
(methods/DKT.py:161-163, methods/DKT_regression.py:53-56).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib

KERNEL_LINEAR = 0
KERNEL_RBF = 1
KERNEL_SQDIST = 2
KERNEL_LINEAR_UNIT = 3      # linear + the promise |a| <= 1 element-wise (rows went through F.normalize)
GRAM_UNIT_ROWS = 1          # the same promise for the rows of Z in gram_bwd
GRAM_W_SYMMETRIC = 2        # gram_bwd: every W[b] is symmetric (as dkt_mll_f32 writes it)
MLL_WANT_GRAD = 1
MLL_WANT_CHOL = 2
MLL_FORCE_GENERIC = 4
MLL_FORCE_REG = 8
MLL_FORCE_BLOCKED = 16
MLL_FORCE_F32MFMA = 32
MLL_E_PER_CLASS = 64
MLL_FORCE_TILED = 128
MLL_FORCE_BAND = 256
MLL_NO_KAPPA_GUARD = 512

LINEAR_KINDS = ("linear", "cossim", "bncossim")
RBF_KINDS = ("rbf", "RBF")
MATERN_KINDS = ("matern",)
POLY_KINDS = {"poli1": 1, "poli2": 2}


def kind_id(kernel: str) -> int:
    if kernel in LINEAR_KINDS:
        return KERNEL_LINEAR
    if kernel in RBF_KINDS:
        return KERNEL_RBF
    raise ValueError("[ERROR] the kernel '" + str(kernel) + "' is not supported!")


def _req(t: torch.Tensor, name: str, ndim: Optional[int] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("dkt_amd.ops: `%s` must be a CUDA/ROCm tensor -- the DKT hot path is HIP-only "
                           "(no CPU fallback)" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("dkt_amd.ops: `%s` must be float32, got %s" % (name, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError("dkt_amd.ops: `%s` must have %d dims, got %s" % (name, ndim, tuple(t.shape)))
    return t.contiguous()


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-kernel timing (bench.py): HIP events recorded on the stream the kernel is launched on
# (torch's current stream), read back after the timed region -- no synchronisation while recording.
_kernel_events = None


def kernel_timing(enable: bool) -> None:
    global _kernel_events
    _kernel_events = {} if enable else None


def kernel_timing_results() -> dict:
    """name -> (launches, mean ms).  Call after torch.cuda.synchronize()."""
    out = {}
    for name, pairs in (_kernel_events or {}).items():
        ms = [a.elapsed_time(b) for a, b in pairs]
        out[name] = (len(ms), sum(ms) / max(len(ms), 1))
    return out


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _kernel_events is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if _kernel_events is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _kernel_events.setdefault(self.name, []).append((self.a, b))
        return False


# ------------------------------------------------------------------------------------------------
# raw (non-differentiable) entry points
# ------------------------------------------------------------------------------------------------
# Switches the PRODUCT library reads (dispatch thresholds / the process-wide exact-fp32 request of include/dkt_abi.h); everything else is a variant switch that only
# the twins library (libdkt_twins.so, -DDKT_TWINS) knows: with DKT_TWINS=1 in the environment AND one of them set, the calls of this module go to that library
# (tests, A/B tools).  Without DKT_TWINS=1 the variant switches have no effect at all.
_PRODUCT_SWITCHES = ("DKT_GRAM_EP_MINB", "DKT_MLL_H2E_MINB", "DKT_MLL_TILED_CHUNK", "DKT_MLL_F32MFMA")
_VARIANT_SWITCHES = ("DKT_GRAM_EP", "DKT_GRAM_SPLIT", "DKT_GRAM_EP_BK", "DKT_GRAM_EP_BD", "DKT_GRAM_UNIT_VAR", "DKT_GRAM_SPLIT_VAR", "DKT_GRAM_BWD_UNIT_VAR",
                     "DKT_GRAM_BWD_SPLIT_VAR", "DKT_GRAM_BWD_UNIT_MIND", "DKT_GRAM_BWD_SPLIT_MIND", "DKT_MLL_TILED_F16", "DKT_GRAM_DIST_EP", "DKT_MLL_P2_GUARD",
                     "DKT_MLL_TILED_WRES", "DKT_MLL_TILED_INVRES", "DKT_GRAM_BIG_EP", "DKT_MLL_TILED_WGS", "DKT_GRAM_BWD_ROWS8", "DKT_MLL_TILED_WDMA", "DKT_CLASS_BWD_V4",
                     "DKT_MLL_TILED_WNW", "DKT_GRAM_SMALL", "DKT_BIG_NB", "DKT_GRAM_BN_F16", "DKT_LDS_STAGE_OLD", "DKT_GRAM_SMALL_WG", "DKT_CLASS_BWD_N128", "DKT_GRAM_FEWEP",
                     "DKT_GRAM_SMALL_XR", "DKT_GRAM_SMALL_LDS")
_ENV_SWITCHES = _PRODUCT_SWITCHES + _VARIANT_SWITCHES
_env_seen = {}


def _sync_env(lib, names=None) -> None:
    """The library reads its switches once; when a test or an A/B tool changed one inside this process, tell it.  `names`: the switches that can matter for this
    library (the product reads the 4 product switches only: ~ 4 instead of ~ 30 environment look-ups per call on the launch-bound small-batch paths)."""
    cur = tuple(os.environ.get(k) for k in (names or _ENV_SWITCHES))
    seen = _env_seen.get(id(lib))
    if cur != seen:
        if seen is not None or any(v is not None for v in cur):
            lib.dkt_reload_env()
        _env_seen[id(lib)] = cur


def _lib_now(want_twin: bool = False):
    """The library this call goes to: the product, or -- DKT_TWINS=1 and a variant switch set (or `want_twin`: a call that names a validation twin the
    product library serves with its generic kernel, e.g. force_f32mfma) -- the twins build of the same ABI.  DKT_TWINS=force: the twins build for every call."""
    tw = os.environ.get("DKT_TWINS")
    if tw is None or tw == "0":                               # the product library, no variant switch can apply: one lookup per call, the 4 product switches synced
        lib = _lib.load()
        _sync_env(lib, _PRODUCT_SWITCHES)
        return lib
    if tw == "force":                                         # (tests of the twins library's own instantiations)
        lib = _lib.load_twins()
    elif os.environ.get("DKT_TWINS") == "1" and (want_twin or os.environ.get("DKT_MLL_F32MFMA") == "1" or any(os.environ.get(k) is not None for k in _VARIANT_SWITCHES)):
        lib = _lib.load_twins()
    else:
        lib = _lib.load()
    _sync_env(lib)
    return lib


def gram(a: torch.Tensor, bm: Optional[torch.Tensor] = None, kind: int = KERNEL_LINEAR,
         lengthscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """E[b] = k(a[b], bm[b]); a:[B,M,D], bm:[B,N,D] or None (symmetric)."""
    a = _req(a, "a", 3)
    b_, m, d = a.shape
    if bm is not None:
        bm = _req(bm, "bm", 3)
        if bm.shape[0] != b_ or bm.shape[2] != d:
            raise RuntimeError("gram: shape mismatch %s vs %s" % (tuple(a.shape), tuple(bm.shape)))
        n = bm.shape[1]
    else:
        n = m
    if kind not in (KERNEL_LINEAR, KERNEL_LINEAR_UNIT):
        lengthscale = _req(lengthscale.reshape(-1), "lengthscale", 1)
    e = torch.empty((b_, m, n), device=a.device, dtype=torch.float32)
    lib = _lib_now()
    with _timed("dkt_gram_f32"):
        st = lib.dkt_gram_f32(_p(a), _p(bm), _p(e), b_, m, n, d, kind, _p(lengthscale), _stream())
    _lib.check(st, "dkt_gram_f32")
    return e


def mll(e: torch.Tensor, y: torch.Tensor, sv: torch.Tensor, mean: torch.Tensor, noise: torch.Tensor,
        want_grad: bool = False, want_chol: bool = False, cls_weight: Optional[torch.Tensor] = None,
        jitter0: float = 1e-6, max_tries: int = 3, force_generic: bool = False, force_reg: bool = False,
        force_blocked: bool = False, force_f32mfma: bool = False, force_tiled: bool = False, force_band: bool = False, no_kappa_guard: bool = False) -> dict:
    """Exact-GP marginal log likelihood of C models per episode.  e:[B,N,N] (one base matrix shared by the class models) or
    [B,C,N,N] (one per class model: DKT_MLL_E_PER_CLASS -- then w is [B,C,N,N] too); y:[C,N] (shared) or [B,C,N].
    force_*: the parity-tested twins of the default kernels (DKT_MLL_FORCE_* of include/dkt_abi.h)."""
    per_class = e.dim() == 4
    e = _req(e, "e", 4 if per_class else 3)
    b_, n, n2 = e.shape[0], e.shape[-2], e.shape[-1]
    if n != n2:
        raise RuntimeError("mll: e must be [B,N,N] or [B,C,N,N]")
    y = _req(y, "y")
    if y.dim() == 2:
        c_, y_bstride = y.shape[0], 0
    elif y.dim() == 3 and y.shape[0] == b_:
        c_, y_bstride = y.shape[1], y.shape[1] * n
    else:
        raise RuntimeError("mll: y must be [C,N] or [B,C,N]")
    if y.shape[-1] != n:
        raise RuntimeError("mll: y last dim %d != N %d" % (y.shape[-1], n))
    sv = _req(sv.reshape(-1), "sv", 1)
    mean = _req(mean.reshape(-1), "mean", 1)
    noise = _req(noise.reshape(-1), "noise", 1)
    if not (sv.numel() == mean.numel() == noise.numel() == c_):
        raise RuntimeError("mll: sv/mean/noise must have C=%d elements" % c_)
    if per_class and e.shape[1] != c_:
        raise RuntimeError("mll: per-class e must be [B,C=%d,N,N]" % c_)
    dev = e.device
    logp = torch.empty((b_, c_), device=dev, dtype=torch.float32)
    alpha = torch.empty((b_, c_, n), device=dev, dtype=torch.float32)
    jit = torch.empty((b_, c_), device=dev, dtype=torch.float32)
    info = torch.empty((b_, c_), device=dev, dtype=torch.int32)
    flags = ((MLL_FORCE_GENERIC if force_generic else 0) | (MLL_FORCE_BLOCKED if force_blocked else 0) |
             (MLL_FORCE_F32MFMA if force_f32mfma else 0) | (MLL_E_PER_CLASS if per_class else 0) | (MLL_FORCE_TILED if force_tiled else 0) | (MLL_FORCE_BAND if force_band else 0) | (MLL_NO_KAPPA_GUARD if no_kappa_guard else 0))
    chol = w = dsv = dmean = dnoise = None
    if want_chol:
        flags |= MLL_WANT_CHOL
        chol = torch.empty((b_, c_, n, n), device=dev, dtype=torch.float32)
    if want_grad:
        flags |= MLL_WANT_GRAD
        w = torch.empty((b_, c_, n, n) if per_class else (b_, n, n), device=dev, dtype=torch.float32)
        dsv = torch.empty((b_, c_), device=dev, dtype=torch.float32)
        dmean = torch.empty((b_, c_), device=dev, dtype=torch.float32)
        dnoise = torch.empty((b_, c_), device=dev, dtype=torch.float32)
    if cls_weight is not None:
        cls_weight = _req(cls_weight.reshape(-1), "cls_weight", 1)
    if force_reg and n + 1 <= 128:                           # (beyond its range the call takes the default kernels, as DKT_MLL_FORCE_REG did)
        # the round-1 register-sweep kernel: a validation twin in the measurement library (libdkt_diag.so) since round 4, not part of the product ABI
        if per_class or force_generic or force_blocked or force_f32mfma:
            raise RuntimeError("mll: force_reg combines with want_grad / want_chol only")
        dlib = _lib.load_diag()
        dlib.dkt_diag_mll_reg_f32.restype = ctypes.c_int
        dlib.dkt_diag_mll_reg_f32.argtypes = ([ctypes.c_void_p] * 2 + [ctypes.c_long] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int, ctypes.c_uint] +
                                              [ctypes.c_void_p] * 11)
        st = dlib.dkt_diag_mll_reg_f32(_p(e), _p(y), y_bstride, _p(sv), _p(mean), _p(noise), b_, c_, n, float(jitter0), int(max_tries),
                                       flags & (MLL_WANT_GRAD | MLL_WANT_CHOL), _p(cls_weight), _p(logp), _p(alpha), _p(chol), _p(w), _p(dsv), _p(dmean), _p(dnoise),
                                       _p(jit), _p(info), _stream())
        _lib.check(st, "dkt_diag_mll_reg_f32")
        return dict(logp=logp, alpha=alpha, jitter=jit, info=info, chol=chol, w=w, dsv=dsv, dmean=dmean, dnoise=dnoise)
    lib = _lib_now(want_twin=force_f32mfma)
    ws_bytes = int(lib.dkt_mll_workspace_bytes_for(b_, c_, n, flags))           # what THIS call needs (the flag-less query covers every flag combination)
    ws = torch.empty((max(ws_bytes, 4) + 3) // 4, device=dev, dtype=torch.float32) if ws_bytes else None
    with _timed("dkt_mll_f32"):
        st = lib.dkt_mll_f32(_p(e), _p(y), y_bstride, _p(sv), _p(mean), _p(noise), b_, c_, n,
                             float(jitter0), int(max_tries), flags, _p(cls_weight), _p(logp), _p(alpha),
                             _p(chol), _p(w), _p(dsv), _p(dmean), _p(dnoise), _p(jit), _p(info), _p(ws),
                             ws_bytes, _stream())
    _lib.check(st, "dkt_mll_f32")
    return dict(logp=logp, alpha=alpha, jitter=jit, info=info, chol=chol, w=w, dsv=dsv, dmean=dmean, dnoise=dnoise)


def gram_bwd(w: torch.Tensor, z: torch.Tensor, ep_scale: Optional[torch.Tensor] = None, unit_rows: bool = False,
             w_symmetric: bool = False) -> torch.Tensor:
    """dZ[b] = ep_scale[b] * (W[b] + W[b]^T) Z[b].  unit_rows: the caller guarantees |z| <= 1 element-wise (rows that went
    through F.normalize), which lets the kernel use the scaled 2-way f16 split (DKT_GRAM_UNIT_ROWS).  w_symmetric: the caller
    states W[b] = W[b]^T (what dkt_mll_f32 writes): DKT_GRAM_W_SYMMETRIC, used by the 128 < N <= 448 kernel."""
    w = _req(w, "w", 3)
    z = _req(z, "z", 3)
    b_, n, d = z.shape
    if tuple(w.shape) != (b_, n, n):
        raise RuntimeError("gram_bwd: w must be [B,N,N]")
    if ep_scale is not None:
        ep_scale = _req(ep_scale.reshape(-1), "ep_scale", 1)
        if ep_scale.numel() != b_:
            raise RuntimeError("gram_bwd: ep_scale must have B elements")
    dz = torch.empty_like(z)
    lib = _lib_now()
    with _timed("dkt_gram_bwd_f32"):
        st = lib.dkt_gram_bwd_f32(_p(w), _p(z), _p(dz), b_, n, d, _p(ep_scale),
                                  (GRAM_UNIT_ROWS if unit_rows else 0) | (GRAM_W_SYMMETRIC if w_symmetric else 0), _stream())
    _lib.check(st, "dkt_gram_bwd_f32")
    return dz


def rbf_bwd(w: torch.Tensor, e: torch.Tensor, lengthscale: torch.Tensor):
    w = _req(w, "w", 3)
    e = _req(e, "e", 3)
    lengthscale = _req(lengthscale.reshape(-1), "lengthscale", 1)
    b_, n, _ = e.shape
    wp = torch.empty_like(e)
    dl = torch.empty((b_,), device=e.device, dtype=torch.float32)
    lib = _lib_now()
    with _timed("dkt_rbf_bwd_f32"):
        st = lib.dkt_rbf_bwd_f32(_p(w), _p(e), _p(lengthscale), _p(wp), _p(dl), b_, n, _stream())
    _lib.check(st, "dkt_rbf_bwd_f32")
    return wp, dl


def sqdist_bwd(w: torch.Tensor, u: torch.Tensor, lengthscale: torch.Tensor):
    w = _req(w, "w", 3)
    u = _req(u, "u", 3)
    lengthscale = _req(lengthscale.reshape(-1), "lengthscale", 1)
    b_, n, _ = u.shape
    wp = torch.empty_like(u)
    dl = torch.empty((b_,), device=u.device, dtype=torch.float32)
    lib = _lib_now()
    _lib.check(lib.dkt_sqdist_bwd_f32(_p(w), _p(u), _p(lengthscale), _p(wp), _p(dl), b_, n, _stream()), "dkt_sqdist_bwd_f32")
    return wp, dl


def _matern25(u: torch.Tensor) -> torch.Tensor:
    """MaternKernel(nu=2.5) from the scaled squared distance (gpytorch clamps d2 >= 1e-30 before the sqrt).  Differentiable torch form, used by
    base_matrix() under autograd; the no-autograd paths take the HIP class map (dkt_class_kernel_f32)."""
    r = torch.sqrt(5.0 * u.clamp_min(1e-30))
    return (1.0 + r + r * r / 3.0) * torch.exp(-r)


def kernel_matrix(a: torch.Tensor, bm: Optional[torch.Tensor], kernel: str, lengthscale: Optional[torch.Tensor] = None,
                  offset: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Base kernel matrix k(a, bm) of ONE model (no autograd): all kernel types of ExactGPLayer (reference DKT.py:352-370).  Matern / polynomial:
    the contraction (dkt_gram_f32) + the element-wise class map of a one-class "episode" (dkt_class_kernel_f32)."""
    if kernel in LINEAR_KINDS:
        return gram(a, bm, KERNEL_LINEAR)
    if kernel in RBF_KINDS:
        return gram(a, bm, KERNEL_RBF, lengthscale)
    if kernel in MATERN_KINDS or kernel in POLY_KINDS:
        for nm, t in (("lengthscale", lengthscale), ("offset", offset)):
            if t is not None and t.numel() != 1:           # ONE model: per-class parameters go to kernel_matrix_per_class (a [C] tensor here would silently mean class 0)
                raise RuntimeError("kernel_matrix: %s must have one element (got %d); use kernel_matrix_per_class for per-class parameters" % (nm, t.numel()))
        return kernel_matrix_per_class(a, bm, kernel, None if lengthscale is None else lengthscale.reshape(-1)[:1],
                                       None if offset is None else offset.reshape(-1)[:1])[:, 0]
    raise ValueError("[ERROR] the kernel '" + str(kernel) + "' is not supported!")


def predict(ex: torch.Tensor, alpha: torch.Tensor, sv: torch.Tensor, mean: torch.Tensor, want_labels: bool = True):
    """mu[b,c,q] = mean[c] + sv[c] sum_n ex[b,(c,)q,n] alpha[b,c,n]; labels[b,q] = argmax_c mu (first maximum wins).
    ex: [B,M,N] (one base cross kernel shared by the class models: dkt_predict_f32) or [B,C,M,N] (one per class model: dkt_predict_per_class_f32)."""
    per_class = ex.dim() == 4
    ex = _req(ex, "ex", 4 if per_class else 3)
    alpha = _req(alpha, "alpha", 3)
    b_, m, n = ex.shape[0], ex.shape[-2], ex.shape[-1]
    c_ = alpha.shape[1]
    if alpha.shape[0] != b_ or alpha.shape[2] != n or (per_class and ex.shape[1] != c_):
        raise RuntimeError("predict: alpha must be [B,C,N] (and a per-class ex [B,C,M,N])")
    sv = _req(sv.reshape(-1), "sv", 1)
    mean = _req(mean.reshape(-1), "mean", 1)
    mu = torch.empty((b_, c_, m), device=ex.device, dtype=torch.float32)
    labels = torch.empty((b_, m), device=ex.device, dtype=torch.int32) if want_labels else None
    lib = _lib_now()
    fn, name = (lib.dkt_predict_per_class_f32, "dkt_predict_per_class_f32") if per_class else (lib.dkt_predict_f32, "dkt_predict_f32")
    _lib.check(fn(_p(ex), _p(alpha), _p(sv), _p(mean), _p(mu), _p(labels), b_, c_, m, n, _stream()), name)
    return mu, labels


def predict_var(ex: torch.Tensor, exx: torch.Tensor, chol: torch.Tensor, sv: torch.Tensor, noise: torch.Tensor):
    ex = _req(ex, "ex", 3)
    exx = _req(exx, "exx", 2)
    chol = _req(chol, "chol", 4)
    b_, m, n = ex.shape
    c_ = chol.shape[1]
    sv = _req(sv.reshape(-1), "sv", 1)
    noise = _req(noise.reshape(-1), "noise", 1)
    var = torch.empty((b_, c_, m), device=ex.device, dtype=torch.float32)
    lib = _lib_now()
    _lib.check(lib.dkt_predict_var_f32(_p(ex), _p(exx), _p(chol), _p(sv), _p(noise), _p(var), b_, c_, m, n, _stream()),
               "dkt_predict_var_f32")
    return var


def smk(x1: torch.Tensor, x2: Optional[torch.Tensor], weights: torch.Tensor, means: torch.Tensor, scales: torch.Tensor,
        want_terms: bool = False):
    """Spectral-mixture matrix E[b] = k(x1[b], x2[b]) (x2 None: symmetric); weights [Q], means / scales [Q,D]
    (constrained values).  Returns (E, Eq) with Eq [B,Q,M,N] the per-mixture terms (None unless want_terms)."""
    x1 = _req(x1, "x1", 3)
    b_, m, d = x1.shape
    if x2 is not None:
        x2 = _req(x2, "x2", 3)
        if x2.shape[0] != b_ or x2.shape[2] != d:
            raise RuntimeError("smk: shape mismatch %s vs %s" % (tuple(x1.shape), tuple(x2.shape)))
        n = x2.shape[1]
    else:
        n = m
    weights = _req(weights.reshape(-1), "weights", 1)
    q = weights.numel()
    means = _req(means.reshape(q, -1), "means", 2)
    scales = _req(scales.reshape(q, -1), "scales", 2)
    if means.shape[1] != d or scales.shape[1] != d:
        raise RuntimeError("smk: means / scales must be [Q,D] with D = %d" % d)
    e = torch.empty((b_, m, n), device=x1.device, dtype=torch.float32)
    eq = torch.empty((b_, q, m, n), device=x1.device, dtype=torch.float32) if want_terms else None
    lib = _lib_now()
    with _timed("dkt_smk_f32"):
        st = lib.dkt_smk_f32(_p(x1), _p(x2), _p(weights), _p(means), _p(scales), _p(e), _p(eq), b_, m, n, d, q, _stream())
    _lib.check(st, "dkt_smk_f32")
    return e, eq


def smk_bwd(ge: torch.Tensor, eq: torch.Tensor, x: torch.Tensor, weights: torch.Tensor, means: torch.Tensor,
            scales: torch.Tensor):
    """Chain rule of the symmetric spectral-mixture matrix: (dx [B,N,D], dmeans [B,Q,D], dscales [B,Q,D])."""
    ge = _req(ge, "ge", 3)
    eq = _req(eq, "eq", 4)
    x = _req(x, "x", 3)
    b_, n, d = x.shape
    q = eq.shape[1]
    weights = _req(weights.reshape(-1), "weights", 1)
    means = _req(means.reshape(q, -1), "means", 2)
    scales = _req(scales.reshape(q, -1), "scales", 2)
    dx = torch.empty_like(x)
    dmeans = torch.empty((b_, q, d), device=x.device, dtype=torch.float32)
    dscales = torch.empty((b_, q, d), device=x.device, dtype=torch.float32)
    lib = _lib_now()
    with _timed("dkt_smk_bwd_f32"):
        st = lib.dkt_smk_bwd_f32(_p(ge), _p(eq), _p(x), _p(weights), _p(means), _p(scales), _p(dx), _p(dmeans), _p(dscales),
                                 b_, n, d, q, _stream())
    _lib.check(st, "dkt_smk_bwd_f32")
    return dx, dmeans, dscales


# ------------------------------------------------------------------------------------------------
# autograd
# ------------------------------------------------------------------------------------------------
class _BaseMatrixFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, lengthscale, kind):
        e = gram(z, None, kind, lengthscale)
        ctx.kind = kind
        ctx.save_for_backward(z, e if kind == KERNEL_RBF else None, lengthscale)
        return e

    @staticmethod
    def backward(ctx, ge):
        z, e, lengthscale = ctx.saved_tensors
        ge = ge.contiguous()
        dl = None
        if ctx.kind == KERNEL_RBF:
            wp, dlb = rbf_bwd(ge, e, lengthscale)
            dz = gram_bwd(wp, z)
            if ctx.needs_input_grad[1]:
                dl = dlb.sum().reshape(lengthscale.shape)
        else:
            dz = gram_bwd(ge, z)
        return (dz if ctx.needs_input_grad[0] else None), dl, None


class _SqDistFn(torch.autograd.Function):
    """U[b] = |z_i - z_j|^2 / l^2 (differentiable in z and l): the building block of the Matern kernel."""

    @staticmethod
    def forward(ctx, z, lengthscale):
        u = gram(z, None, KERNEL_SQDIST, lengthscale)
        ctx.save_for_backward(z, u, lengthscale)
        return u

    @staticmethod
    def backward(ctx, gu):
        z, u, lengthscale = ctx.saved_tensors
        wp, dlb = sqdist_bwd(gu.contiguous(), u, lengthscale)
        dz = gram_bwd(wp, z) if ctx.needs_input_grad[0] else None
        dl = dlb.sum().reshape(lengthscale.shape) if ctx.needs_input_grad[1] else None
        return dz, dl


class _SpectralMixtureFn(torch.autograd.Function):
    """E = SpectralMixtureKernel(z, z) differentiable in z, the mixture weights, means and scales."""

    @staticmethod
    def forward(ctx, z, weights, means, scales):
        e, eq = smk(z, None, weights, means, scales, want_terms=True)
        ctx.save_for_backward(z, eq, weights, means, scales)
        return e

    @staticmethod
    def backward(ctx, ge):
        z, eq, weights, means, scales = ctx.saved_tensors
        ge = ge.contiguous()
        dz, dmb, dsb = smk_bwd(ge, eq, z, weights, means, scales)
        dw = (ge.unsqueeze(1) * eq).sum((0, 2, 3)).reshape(weights.shape) if ctx.needs_input_grad[1] else None
        dm = dmb.sum(0).reshape(means.shape) if ctx.needs_input_grad[2] else None
        ds = dsb.sum(0).reshape(scales.shape) if ctx.needs_input_grad[3] else None
        return (dz if ctx.needs_input_grad[0] else None), dw, dm, ds


def spectral_mixture_matrix(z: torch.Tensor, weights: torch.Tensor, means: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """Differentiable symmetric spectral-mixture matrix [B,N,N] of z [B,N,D] (reference DKT_regression.py:121-122)."""
    return _SpectralMixtureFn.apply(z, weights, means, scales)


CLASSMAP_RBF, CLASSMAP_MATERN25, CLASSMAP_POLY = 0, 1, 2


def _classmap_of(kernel: str, lengthscale, offset):
    """(map id, power, parameter [C], base-matrix kind) of a kernel whose class models own their base-kernel parameter."""
    if kernel in POLY_KINDS:
        return CLASSMAP_POLY, POLY_KINDS[kernel], offset, KERNEL_LINEAR
    if kernel in MATERN_KINDS:
        return CLASSMAP_MATERN25, 0, lengthscale, KERNEL_SQDIST
    if kernel in RBF_KINDS:
        return CLASSMAP_RBF, 0, lengthscale, KERNEL_SQDIST
    raise ValueError("[ERROR] the kernel '" + str(kernel) + "' has no per-class base-kernel parameter")


def class_kernel(base: torch.Tensor, cmap: int, power: int, param: torch.Tensor) -> torch.Tensor:
    """E[B,C,...] = f(base[B,...]; param[c]) -- the C class kernels of an episode from its ONE contraction (dkt_class_kernel_f32)."""
    base = _req(base, "base")
    param = _req(param.reshape(-1), "param", 1)
    b_, c = base.shape[0], param.numel()
    nn = base[0].numel()
    e = torch.empty((b_, c) + tuple(base.shape[1:]), device=base.device, dtype=torch.float32)
    lib = _lib_now()
    with _timed("dkt_class_kernel_f32"):
        st = lib.dkt_class_kernel_f32(_p(base), int(cmap), _p(param), int(power), _p(e), b_, c, nn, _stream())
    _lib.check(st, "dkt_class_kernel_f32")
    return e


def class_kernel_bwd(w: torch.Tensor, base: torch.Tensor, cmap: int, power: int, param: torch.Tensor):
    """Chain rule behind the per-class marginal-likelihood launch: w [B,C,N,N] = d obj / d E (symmetric per matrix) -> (Wp [B,N,N] for
    gram_bwd, dparam [B,C])."""
    w = _req(w, "w", 4)
    base = _req(base, "base", 3)
    param = _req(param.reshape(-1), "param", 1)
    b_, c, n, _ = w.shape
    if tuple(base.shape) != (b_, n, n) or param.numel() != c:
        raise RuntimeError("class_kernel_bwd: base must be [B,N,N] and param [C]")
    wp = torch.empty_like(base)
    lib = _lib_now()
    nsplit = int(lib.dkt_class_kernel_bwd_nsplit(b_, n))
    dparam = torch.empty((b_, nsplit, c), device=w.device, dtype=torch.float32)
    with _timed("dkt_class_kernel_bwd_f32"):
        st = lib.dkt_class_kernel_bwd_f32(_p(w), _p(base), int(cmap), _p(param), int(power), _p(wp), _p(dparam), b_, c, n, _stream())
    _lib.check(st, "dkt_class_kernel_bwd_f32")
    return wp, (dparam[:, 0] if nsplit == 1 else dparam.sum(1))


class _EpisodeLossClassKernelFn(torch.autograd.Function):
    """Fused training episode for the kernels whose class models own a base-kernel parameter (rbf / matern: lengthscale [C]; poli1 / poli2:
    offset [C]; reference DKT.py:63-66, 352-365):
       forward : ONE contraction per episode (dkt_gram_f32: squared distances or Gram) -> the C class kernels (dkt_class_kernel_f32)
                 -> logp, W[B,C,N,N], hyper grads in ONE launch (dkt_mll_f32, DKT_MLL_E_PER_CLASS)
       backward: dkt_class_kernel_bwd_f32 (sum over the classes, parameter gradients) -> dkt_gram_bwd_f32 (upstream grad as ep_scale)."""

    @staticmethod
    def forward(ctx, z, y, sv, mean, noise, cls_weight, param, cmap, power, base_kind, jitter0, max_tries):
        one = torch.ones(1, device=z.device, dtype=torch.float32)
        base = gram(z, None, base_kind, one if base_kind == KERNEL_SQDIST else None)
        e = class_kernel(base, cmap, power, param)
        out = mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cls_weight, jitter0=jitter0, max_tries=max_tries)
        obj = objective(out["logp"], cls_weight)
        ctx.save_for_backward(z, base, out["w"], param, out["dsv"], out["dmean"], out["dnoise"], cls_weight)
        ctx.maps = (int(cmap), int(power))
        ctx.shapes = (sv.shape, mean.shape, noise.shape, param.shape)
        ctx.mark_non_differentiable(out["logp"], out["alpha"], out["info"], out["jitter"], e)
        ctx.set_materialize_grads(False)
        return obj, out["logp"], out["alpha"], out["info"], out["jitter"], e

    @staticmethod
    def backward(ctx, gobj, *_unused):
        if gobj is None:
            return (None,) * 12
        z, base, w, param, dsv, dmean, dnoise, cw = ctx.saved_tensors
        gobj = gobj.contiguous()
        ng = ctx.needs_input_grad
        dz = gparam = None
        if ng[0] or ng[6]:
            wp, dpar = class_kernel_bwd(w, base, ctx.maps[0], ctx.maps[1], param)
            if ng[0]:
                dz = gram_bwd(wp, z, gobj)
            if ng[6]:
                gparam = (gobj.reshape(-1, 1) * dpar).sum(0).reshape(ctx.shapes[3])
        gsv, gmean, gnoise = hyper_grads(gobj, cw, dsv if ng[2] else None, dmean if ng[3] else None, dnoise if ng[4] else None, ctx.shapes[:3])
        return dz, None, gsv, gmean, gnoise, None, gparam, None, None, None, None, None


def mll_per_class_supported(n: int, c: int) -> bool:
    """Sizes the one-launch per-class path serves: dkt_mll_f32 with DKT_MLL_E_PER_CLASS takes every N in one call (N <= 127 one wave per matrix,
    128 <= N <= 447 the tile-array pipeline; both with the jitter ladder); beyond N = 447 its one-launch kernel is the generic one, and the host
    prefers one single-model call per class there (the blocked path serves those).  dkt_class_kernel_bwd_f32: C <= 32."""
    return c <= 32 and n + 1 <= 448


def episode_loss_class_kernel(z, y, sv, mean, noise, cls_weight, kernel: str, lengthscale=None, offset=None,
                              jitter0: float = 1e-6, max_tries: int = 3):
    """Training episode(s) z:[B,N,D] for rbf / matern / poli1 / poli2 with per-class lengthscale / offset [C], sizes of
    mll_per_class_supported().
    Returns (obj[B], logp[B,C], alpha[B,C,N], info[B,C], jitter[B,C], E[B,C,N,N])."""
    cmap, power, param, base_kind = _classmap_of(kernel, lengthscale, offset)
    return _EpisodeLossClassKernelFn.apply(_req(z, "z", 3), y, sv, mean, noise, cls_weight, param, cmap, power, base_kind, jitter0, max_tries)


def base_matrix_per_class(z: torch.Tensor, kernel: str, lengthscale: Optional[torch.Tensor] = None,
                          offset: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable E[B,C,N,N] for the kernels whose class models own their base-kernel parameter (one ExactGPLayer per class,
    reference DKT.py:63-66, 352-365): rbf / matern (lengthscale [C]), poli1 / poli2 (offset [C]).  The O(N^2 D) contraction -- the
    squared distances or the Gram -- is built ONCE per episode (dkt_gram_f32); the per-class map is element-wise."""
    one = torch.ones(1, device=z.device, dtype=torch.float32)
    if kernel in POLY_KINDS:
        g = _BaseMatrixFn.apply(z, torch.zeros(1, device=z.device, dtype=torch.float32), KERNEL_LINEAR)
        return (g.unsqueeze(1) + offset.reshape(1, -1, 1, 1)) ** POLY_KINDS[kernel]
    u = _SqDistFn.apply(z, one).unsqueeze(1) / (lengthscale.reshape(1, -1, 1, 1) ** 2)         # |z_i - z_j|^2 / l_c^2
    if kernel in MATERN_KINDS:
        return _matern25(u)
    if kernel in RBF_KINDS:
        return torch.exp(-0.5 * u)
    raise ValueError("[ERROR] the kernel '" + str(kernel) + "' has no per-class base-kernel parameter")


def kernel_matrix_per_class(a: torch.Tensor, bm: Optional[torch.Tensor], kernel: str, lengthscale: Optional[torch.Tensor] = None,
                            offset: Optional[torch.Tensor] = None) -> torch.Tensor:
    """k_c(a, bm) for every class model, [B,C,M,N] (no autograd): one contraction (dkt_gram_f32), the C class maps in one launch
    (dkt_class_kernel_f32)."""
    cmap, power, param, base_kind = _classmap_of(kernel, lengthscale, offset)
    one = torch.ones(1, device=a.device, dtype=torch.float32)
    return class_kernel(gram(a, bm, base_kind, one if base_kind == KERNEL_SQDIST else None), cmap, power, param.detach())


def base_matrix(z: torch.Tensor, kernel: str = "bncossim", lengthscale: Optional[torch.Tensor] = None,
                offset: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable symmetric base kernel matrix E[B,N,N] of z[B,N,D] for every kernel type of the reference's
    ExactGPLayer (DKT.py:352-370): linear / cossim / bncossim, rbf, matern (nu = 2.5), poli1, poli2."""
    if kernel in MATERN_KINDS:
        return _matern25(_SqDistFn.apply(z, lengthscale))
    if kernel in POLY_KINDS:
        g = _BaseMatrixFn.apply(z, torch.zeros(1, device=z.device, dtype=torch.float32), KERNEL_LINEAR)
        return (g + offset.reshape(())) ** POLY_KINDS[kernel]
    kind = kind_id(kernel)
    if kind == KERNEL_RBF and lengthscale is None:
        raise RuntimeError("rbf needs a lengthscale tensor")
    if lengthscale is None:
        lengthscale = torch.zeros(1, device=z.device, dtype=torch.float32)
    return _BaseMatrixFn.apply(z, lengthscale, kind)


class _MllObjectiveFn(torch.autograd.Function):
    """obj[b] = sum_c cls_weight[c] logp[b,c]; gradients were produced by the same kernel launch."""

    @staticmethod
    def forward(ctx, e, y, sv, mean, noise, cls_weight, jitter0, max_tries):
        out = mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cls_weight, jitter0=jitter0, max_tries=max_tries)
        obj = objective(out["logp"], cls_weight)
        ctx.save_for_backward(out["w"], out["dsv"], out["dmean"], out["dnoise"], cls_weight)
        ctx.shapes = (sv.shape, mean.shape, noise.shape)
        ctx.mark_non_differentiable(out["logp"], out["alpha"], out["info"], out["jitter"])
        ctx.set_materialize_grads(False)       # (otherwise autograd zero-fills a gradient tensor for every non-differentiable output)
        return obj, out["logp"], out["alpha"], out["info"], out["jitter"]

    @staticmethod
    def backward(ctx, gobj, *_unused):
        if gobj is None:
            return (None,) * 8
        w, dsv, dmean, dnoise, cw = ctx.saved_tensors
        gobj = gobj.contiguous()
        ge = w * gobj.reshape([-1] + [1] * (w.dim() - 1)) if ctx.needs_input_grad[0] else None      # w: [B,N,N] or [B,C,N,N] (per-class E)
        gsv, gmean, gnoise = hyper_grads(gobj, cw, dsv if ctx.needs_input_grad[2] else None, dmean if ctx.needs_input_grad[3] else None, dnoise if ctx.needs_input_grad[4] else None, ctx.shapes[:3])
        return ge, None, gsv, gmean, gnoise, None, None, None


def objective(logp: torch.Tensor, cls_weight: Optional[torch.Tensor]) -> torch.Tensor:
    """obj[b] = sum_c cls_weight[c] logp[b,c] (dkt_objective_f32: one launch, fixed order) -- the sum over the class models of SumMarginalLogLikelihood (DKT.py:70-71, 161)."""
    if os.environ.get("DKT_FUSED_REDUCTIONS", "1") == "0":                 # the tensor expressions (their twin; three launches)
        return logp.sum(1) if cls_weight is None else (logp * cls_weight.reshape(1, -1)).sum(1)
    logp = _req(logp, "logp", 2)
    b_, c_ = logp.shape
    cw = None if cls_weight is None else _req(cls_weight.reshape(-1), "cls_weight", 1)
    obj = torch.empty((b_,), device=logp.device, dtype=torch.float32)
    with _timed("dkt_objective_f32"):
        _lib.check(_lib_now().dkt_objective_f32(_p(logp), _p(cw), _p(obj), b_, c_, _stream()), "dkt_objective_f32")
    return obj


def hyper_grads(gobj: torch.Tensor, cls_weight: Optional[torch.Tensor], dsv, dmean, dnoise, shapes):
    """(g_sv, g_mean, g_noise)[c] = cls_weight[c] sum_b gobj[b] d_x[b,c] for the d_x that are not None (dkt_hyper_grads_f32: ONE launch, fixed summation order),
    reshaped to `shapes`; None where d_x is None."""
    if os.environ.get("DKT_FUSED_REDUCTIONS", "1") == "0":                 # the tensor expressions (their twin; a multiply + two launches per parameter)
        gw = gobj.reshape(-1, 1) * (1.0 if cls_weight is None else cls_weight.reshape(1, -1))
        return tuple(None if d is None else (gw * d).sum(0).reshape(sh) for d, sh in zip((dsv, dmean, dnoise), shapes))
    gobj = _req(gobj.reshape(-1), "gobj", 1)
    ds = [None if d is None else _req(d, "d", 2) for d in (dsv, dmean, dnoise)]
    if all(d is None for d in ds):
        return None, None, None
    b_, c_ = next(d for d in ds if d is not None).shape
    cw = None if cls_weight is None else _req(cls_weight.reshape(-1), "cls_weight", 1)
    gs = [None if d is None else torch.empty((c_,), device=gobj.device, dtype=torch.float32) for d in ds]
    with _timed("dkt_hyper_grads_f32"):
        _lib.check(_lib_now().dkt_hyper_grads_f32(_p(gobj), _p(cw), _p(ds[0]), _p(ds[1]), _p(ds[2]), _p(gs[0]), _p(gs[1]), _p(gs[2]), b_, c_, _stream()), "dkt_hyper_grads_f32")
    return tuple(None if g is None else g.reshape(sh) for g, sh in zip(gs, shapes))


def mll_objective(e, y, sv, mean, noise, cls_weight, jitter0: float = 1e-6, max_tries: int = 3):
    """Returns (obj[B], logp[B,C], alpha[B,C,N], info[B,C], jitter[B,C])."""
    return _MllObjectiveFn.apply(e, y, sv, mean, noise, cls_weight, jitter0, max_tries)


class _EpisodeLossLinearFn(torch.autograd.Function):
    """Fused training episode for the linear / cossim / bncossim kernel:
       forward : E = Z Z^T (dkt_gram_f32) -> logp, W, hyper grads (dkt_mll_f32, one launch)
       backward: dZ = g_b (W + W^T) Z (dkt_gram_bwd_f32, upstream grad folded in as ep_scale)."""

    @staticmethod
    def forward(ctx, z, y, sv, mean, noise, cls_weight, jitter0, max_tries, unit_rows=False):
        e = gram(z, None, KERNEL_LINEAR_UNIT if unit_rows else KERNEL_LINEAR)
        out = mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cls_weight, jitter0=jitter0, max_tries=max_tries)
        obj = objective(out["logp"], cls_weight)
        ctx.save_for_backward(z, out["w"], out["dsv"], out["dmean"], out["dnoise"], cls_weight)
        ctx.shapes = (sv.shape, mean.shape, noise.shape)
        ctx.unit_rows = bool(unit_rows)
        ctx.mark_non_differentiable(out["logp"], out["alpha"], out["info"], out["jitter"], e)
        ctx.set_materialize_grads(False)       # (otherwise autograd zero-fills a gradient tensor for every non-differentiable output: E alone is 361 MB at cfg2)
        return obj, out["logp"], out["alpha"], out["info"], out["jitter"], e

    @staticmethod
    def backward(ctx, gobj, *_unused):
        if gobj is None:
            return (None,) * 9
        z, w, dsv, dmean, dnoise, cw = ctx.saved_tensors
        gobj = gobj.contiguous()
        dz = gram_bwd(w, z, gobj, unit_rows=ctx.unit_rows, w_symmetric=True) if ctx.needs_input_grad[0] else None   # W: from dkt_mll_f32
        gsv, gmean, gnoise = hyper_grads(gobj, cw, dsv if ctx.needs_input_grad[2] else None, dmean if ctx.needs_input_grad[3] else None, dnoise if ctx.needs_input_grad[4] else None, ctx.shapes[:3])
        return dz, None, gsv, gmean, gnoise, None, None, None, None


# ------------------------------------------------------------------------------------------------------
# Linear kernels in FEATURE space: D <= 64 < N (Conv4S / Omniglot; include/dkt_abi.h "dkt_lowrank_*", csrc/dkt_lowrank.hip)
# ------------------------------------------------------------------------------------------------------
FUSED_EP_MAX_N = 128          # the episode-resident fused kernels of dkt_frontend.hip (Zn never written); above it: dkt_frontend_big.hip
LOWRANK_DP = 64               # DKT_LOWRANK_DP
LOWRANK_MIN_N = 80            # below it the D x D problem (5 x 5 tiles) is no smaller than the N x N one
LOWRANK_MIN_B = 3072          # episodes of N <= 128 rows: batches from which the feature-space step wins (below, both steps are bound by their launches, and it has 5 to the N x N step's 3:
                              # tools/lowrank_small_batch_probe.py -- 0.35 vs 0.31 ms up to 1024 episodes of 105 rows; 0.69 vs 1.03 ms at 8192)


def lowrank_supported(n: int, d: int, c: int) -> bool:
    """Shapes the feature-space calls (dkt_lowrank_*) serve and are meant for: K_c = sv_c Z Z^T + noise_c I with D <= 64, D % 4 == 0, C <= 32 and an episode of at
    least 80 rows (the Omniglot / Conv4S episodes: D = 64, N = 105 or 420)."""
    return d <= LOWRANK_DP and d % 4 == 0 and c <= 32 and n >= LOWRANK_MIN_N


def lowrank_applies(n: int, d: int, c: int, b: int = LOWRANK_MIN_B, front_end: bool = False) -> bool:
    """Does this call take the feature-space episode?  DKT_LOWRANK: 1 (default) = where it wins -- every batch of episodes with more than 128 rows (one 420-row episode:
    0.35 instead of 1.2 ms, 1024 of them: 0.49 instead of 16.9 ms), and for N <= 128 from LOWRANK_MIN_B episodes per call (never behind the fused bn_out front end, whose
    episode-resident kernels for N <= 128 are as fast at 8192 episodes and faster below); 0 = never (the N x N kernels: the twin the tests compare); force = wherever the
    shape is supported (tests)."""
    mode = os.environ.get("DKT_LOWRANK", "1")
    if mode == "0" or not lowrank_supported(n, d, c):
        return False
    if mode == "force" or n > FUSED_EP_MAX_N:
        return True
    return (not front_end) and b >= LOWRANK_MIN_B


_lowrank_zeros = {}


def _zeros_cached(c: int, dev) -> torch.Tensor:
    """[C] zeros (the mean of the D x D models), one tensor per (device, C): never written."""
    if torch.cuda.is_current_stream_capturing():          # (a tensor born inside a graph capture lives in the graph's private pool: not cached)
        return torch.zeros(c, device=dev, dtype=torch.float32)
    key = (dev, int(c))
    t = _lowrank_zeros.get(key)
    if t is None:
        t = _lowrank_zeros[key] = torch.zeros(c, device=dev, dtype=torch.float32)
    return t


def _lowrank_forward(z, y, sv_, mean_, noise_in, cw_, jitter0, max_tries, unit_rows) -> dict:
    """The forward calls of the feature-space episode on contiguous tensors (no autograd): noise floor -> A, P -> the D x D models -> alpha, logp, gradients.
    Returns obj, logp, alpha, info, jitter (total), v / t / wd (for the backward), dsv, dmean, dnoise."""
    b_, n, d = z.shape
    if y.dim() == 2:
        c_, y_bstride = y.shape[0], 0
    elif y.dim() == 3 and y.shape[0] == b_:
        c_, y_bstride = y.shape[1], y.shape[1] * n
    else:
        raise RuntimeError("episode_loss_linear: y must be [C,N] or [B,C,N]")
    dev = z.device
    lib = _lib_now()
    # the rung of the jitter ladder that lifts the noise floor of the (rank-deficient) N x N matrix above fp32 rounding -- 0 for any sane noise; rows that
    # went through F.normalize have |z_i|^2 = 1, otherwise one reduction over Z finds the largest diagonal element of Z Z^T.  NOTE: that
    # maximum is taken over the WHOLE call, so for un-normalised `linear` features with a noise below 2^-22 max K_ii one large-norm row raises the rung of every
    # episode of the batch (the N x N path and psd_safe_cholesky decide per matrix).  cossim / bncossim (unit rows) and any noise >= 1e-5 are unaffected; the rung is
    # reported per (episode, class) in `jitter` either way.
    zmax2 = None if unit_rows else z.square().sum(2).amax().reshape(1).contiguous()
    noise_ = torch.empty_like(noise_in)
    pre = torch.empty_like(noise_in)
    _lib.check(lib.dkt_lowrank_noise_floor_f32(_p(sv_), _p(noise_in), _p(zmax2), float(jitter0), int(max_tries), _p(noise_), _p(pre), c_, _stream()),
               "dkt_lowrank_noise_floor_f32")
    a = torch.empty((b_, LOWRANK_DP, LOWRANK_DP), device=dev, dtype=torch.float32)
    p = torch.empty((b_, c_, LOWRANK_DP), device=dev, dtype=torch.float32)
    with _timed("dkt_lowrank_gram_f32"):
        st = lib.dkt_lowrank_gram_f32(_p(z), _p(y), y_bstride, _p(mean_), _p(a), _p(p), b_, c_, n, d, _stream())
    _lib.check(st, "dkt_lowrank_gram_f32")
    # the D x D models.  Unit rows (cossim / bncossim): cond(K') <= 1 + sv lambda_max / noise with lambda_max far below the trace N the kappa guard of dkt_mll_f32 prices
    # (N = 420 rows would trip it at sv > 1.2 for a true condition of a few hundred); rows without the promise keep the guard
    out = mll(a, p, sv_, _zeros_cached(c_, dev), noise_, want_grad=True, cls_weight=cw_, jitter0=jitter0, max_tries=max_tries, no_kappa_guard=bool(unit_rows))
    logp = torch.empty((b_, c_), device=dev, dtype=torch.float32)
    alpha = torch.empty((b_, c_, n), device=dev, dtype=torch.float32)
    v = torch.empty((b_, c_, n), device=dev, dtype=torch.float32)
    dsv = torch.empty((b_, c_), device=dev, dtype=torch.float32)
    dmean = torch.empty((b_, c_), device=dev, dtype=torch.float32)
    dnoise = torch.empty((b_, c_), device=dev, dtype=torch.float32)
    jit = torch.empty((b_, c_), device=dev, dtype=torch.float32)
    obj = torch.empty((b_,), device=dev, dtype=torch.float32)
    with _timed("dkt_lowrank_finish_f32"):
        st = lib.dkt_lowrank_finish_f32(_p(z), _p(y), y_bstride, _p(sv_), _p(mean_), _p(noise_), _p(cw_), _p(out["alpha"]), _p(out["logp"]), _p(out["dnoise"]),
                                        _p(out["jitter"]), _p(pre), _p(jit), _p(obj), _p(logp), _p(alpha), _p(v), _p(dsv), _p(dmean), _p(dnoise), b_, c_, n, d, _stream())
    _lib.check(st, "dkt_lowrank_finish_f32")
    return dict(obj=obj, logp=logp, alpha=alpha, info=out["info"], jitter=jit, v=v, t=out["alpha"], wd=out["w"], dsv=dsv, dmean=dmean, dnoise=dnoise)


def _lowrank_backward(z, v, t, wd, gobj) -> torch.Tensor:
    """dZ[b] = gobj[b] (V^T T + 2 Z W')  (dkt_lowrank_bwd_f32)."""
    b_, n, d = z.shape
    dz = torch.empty_like(z)
    lib = _lib_now()
    with _timed("dkt_lowrank_bwd_f32"):
        st = lib.dkt_lowrank_bwd_f32(_p(z), _p(v), _p(t), _p(wd), _p(_req(gobj.reshape(-1), "gobj", 1)), _p(dz), b_, v.shape[1], n, d, _stream())
    _lib.check(st, "dkt_lowrank_bwd_f32")
    return dz


class _EpisodeLossLowRankFn(torch.autograd.Function):
    """Training episode of the linear / cossim / bncossim kernel in feature space (D <= 64 < N):
       forward : A = Z^T Z, P = Z^T (Y - m) (dkt_lowrank_gram_f32) -> the D x D model K'_c = sv_c A + noise_c I through dkt_mll_f32 (jitter ladder and all)
                 -> alpha, logp, hyper-parameter gradients, V (dkt_lowrank_finish_f32)
       backward: dZ = g_b (V^T T + 2 Z W') (dkt_lowrank_bwd_f32).
    Neither E[B,N,N] nor W[B,N,N] is ever formed (reference lines replaced: methods/DKT.py:375-378, 161-163)."""

    @staticmethod
    def forward(ctx, z, y, sv, mean, noise, cls_weight, jitter0, max_tries, unit_rows):
        z = _req(z, "z", 3)
        o = _lowrank_forward(z, _req(y, "y"), _req(sv.reshape(-1), "sv", 1), _req(mean.reshape(-1), "mean", 1), _req(noise.reshape(-1), "noise", 1),
                             _req(cls_weight.reshape(-1), "cls_weight", 1), jitter0, max_tries, unit_rows)
        ctx.save_for_backward(z, o["v"], o["t"], o["wd"], o["dsv"], o["dmean"], o["dnoise"], _req(cls_weight.reshape(-1), "cls_weight", 1))
        ctx.shapes = (sv.shape, mean.shape, noise.shape)
        ctx.mark_non_differentiable(o["logp"], o["alpha"], o["info"], o["jitter"])
        ctx.set_materialize_grads(False)
        return o["obj"], o["logp"], o["alpha"], o["info"], o["jitter"]

    @staticmethod
    def backward(ctx, gobj, *_unused):
        if gobj is None:
            return (None,) * 9
        z, v, t, wd, dsv, dmean, dnoise, cw = ctx.saved_tensors
        gobj = gobj.contiguous()
        dz = _lowrank_backward(z, v, t, wd, gobj) if ctx.needs_input_grad[0] else None
        gsv, gmean, gnoise = hyper_grads(gobj, cw, dsv if ctx.needs_input_grad[2] else None, dmean if ctx.needs_input_grad[3] else None, dnoise if ctx.needs_input_grad[4] else None, ctx.shapes[:3])
        return dz, None, gsv, gmean, gnoise, None, None, None, None


# ------------------------------------------------------------------------------------------------------
# BNCosSim front half fused into the Gram build (reference methods/DKT.py:48,141-142,375-378)
# ------------------------------------------------------------------------------------------------------
def bn_stats(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float = 1e-5) -> dict:
    """Train-mode BatchNorm1d statistics per episode: x:[B,N,D] -> mean, rstd, a, s, var_unbiased (each [B,D]);
    y = a x + s is bn_out(x)."""
    x = _req(x, "x", 3)
    b_, n, d = x.shape
    if d % 4:
        raise RuntimeError("bn_stats: D must be a multiple of 4")
    gamma = None if gamma is None else _req(gamma.reshape(-1), "gamma", 1)
    beta = None if beta is None else _req(beta.reshape(-1), "beta", 1)
    out = {k: torch.empty((b_, d), device=x.device, dtype=torch.float32) for k in ("mean", "rstd", "a", "s", "var_unbiased")}
    lib = _lib_now()
    with _timed("dkt_bn_stats_f32"):
        st = lib.dkt_bn_stats_f32(_p(x), _p(gamma), _p(beta), float(eps), _p(out["mean"]), _p(out["rstd"]), _p(out["a"]),
                                  _p(out["s"]), _p(out["var_unbiased"]), b_, n, d, _stream())
    _lib.check(st, "dkt_bn_stats_f32")
    return out


def _ab_stride(a: torch.Tensor, s: torch.Tensor, b_: int, d: int) -> int:
    if a.shape != s.shape:
        raise RuntimeError("gram_bn: a and s must have the same shape")
    if a.dim() == 1 and a.shape[0] == d:
        return 0
    if a.dim() == 2 and tuple(a.shape) == (b_, d):
        return d
    raise RuntimeError("gram_bn: a / s must be [D] or [B,D]")


def gram_bn(x: torch.Tensor, a: torch.Tensor, s: torch.Tensor):
    """E[b] = Zn Zn^T with Zn = normalize(a x + s) (never materialised); returns (E [B,N,N], rnorm [B,N])."""
    x = _req(x, "x", 3)
    b_, n, d = x.shape
    if n > 128 or d % 4:
        raise RuntimeError("gram_bn: needs N <= 128 and D %% 4 == 0 (got N=%d, D=%d)" % (n, d))
    a = _req(a, "a")
    s = _req(s, "s")
    stride = _ab_stride(a, s, b_, d)
    e = torch.empty((b_, n, n), device=x.device, dtype=torch.float32)
    rnorm = torch.empty((b_, n), device=x.device, dtype=torch.float32)
    lib = _lib_now()
    with _timed("dkt_gram_bn_f32"):
        st = lib.dkt_gram_bn_f32(_p(x), _p(a), _p(s), stride, _p(e), _p(rnorm), b_, n, d, _stream())
    _lib.check(st, "dkt_gram_bn_f32")
    return e, rnorm


def gram_bn_train(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float = 1e-5):
    """bn_stats + gram_bn in one pass over x (train-mode BatchNorm1d statistics taken inside the Gram kernel's staging path).
    Returns (E [B,N,N], rnorm [B,N], stats dict as bn_stats)."""
    x = _req(x, "x", 3)
    b_, n, d = x.shape
    if n > 128 or d % 4:
        raise RuntimeError("gram_bn_train: needs N <= 128 and D %% 4 == 0 (got N=%d, D=%d)" % (n, d))
    gamma = None if gamma is None else _req(gamma.reshape(-1), "gamma", 1)
    beta = None if beta is None else _req(beta.reshape(-1), "beta", 1)
    out = {k: torch.empty((b_, d), device=x.device, dtype=torch.float32) for k in ("mean", "rstd", "a", "s", "var_unbiased")}
    e = torch.empty((b_, n, n), device=x.device, dtype=torch.float32)
    rnorm = torch.empty((b_, n), device=x.device, dtype=torch.float32)
    lib = _lib_now()
    with _timed("dkt_gram_bn_train_f32"):
        st = lib.dkt_gram_bn_train_f32(_p(x), _p(gamma), _p(beta), float(eps), _p(out["mean"]), _p(out["rstd"]), _p(out["a"]),
                                       _p(out["s"]), _p(out["var_unbiased"]), _p(e), _p(rnorm), b_, n, d, _stream())
    _lib.check(st, "dkt_gram_bn_train_f32")
    return e, rnorm, out


def gram_bn_bwd(w, e, x, a, s, rnorm, mean=None, rstd=None, ep_scale=None):
    """Backward of gram_bn (+ train-mode batch statistics when mean/rstd are given): returns (dX, dgamma_part, dbeta_part);
    the last two are [B,D] per-episode parts (None in eval mode)."""
    x = _req(x, "x", 3)
    b_, n, d = x.shape
    w = _req(w, "w", 3)
    e = _req(e, "e", 3)
    a = _req(a, "a")
    s = _req(s, "s")
    rnorm = _req(rnorm, "rnorm", 2)
    stride = _ab_stride(a, s, b_, d)
    train = mean is not None
    if train:
        mean = _req(mean, "mean", 2)
        rstd = _req(rstd, "rstd", 2)
    if ep_scale is not None:
        ep_scale = _req(ep_scale.reshape(-1), "ep_scale", 1)
    dx = torch.empty_like(x)
    dg = torch.empty((b_, d), device=x.device, dtype=torch.float32) if train else None
    db = torch.empty((b_, d), device=x.device, dtype=torch.float32) if train else None
    lib = _lib_now()
    with _timed("dkt_gram_bn_bwd_f32"):
        st = lib.dkt_gram_bn_bwd_f32(_p(w), _p(e), _p(x), _p(a), _p(s), stride, _p(mean), _p(rstd), _p(rnorm), _p(ep_scale),
                                     _p(dx), _p(dg), _p(db), b_, n, d, _stream())
    _lib.check(st, "dkt_gram_bn_bwd_f32")
    return dx, dg, db


def bn_param_grads(dg: torch.Tensor, db: torch.Tensor):
    """(dgamma [D], dbeta [D]) = the sums over the episodes of the per-episode parts [B,D] gram_bn_bwd / normalize_bn_bwd return (dkt_bn_param_grads_f32: fixed
    order; one launch up to 256 episodes, two beyond -- as two tensor reductions: two launches + four buffer fills).  One episode: the parts ARE the sums."""
    dg = _req(dg, "dgamma_part", 2)
    db = _req(db, "dbeta_part", 2)
    b_, d = dg.shape
    if db.shape != dg.shape:
        raise RuntimeError("bn_param_grads: the two parts must have the same shape")
    if b_ == 1:
        return dg.reshape(d), db.reshape(d)
    if os.environ.get("DKT_FUSED_REDUCTIONS", "1") == "0":                 # the tensor expressions (their twin)
        return dg.sum(0), db.sum(0)
    lib = _lib_now()
    og = torch.empty((d,), device=dg.device, dtype=torch.float32)
    ob = torch.empty((d,), device=dg.device, dtype=torch.float32)
    ws_bytes = int(lib.dkt_bn_param_grads_workspace_bytes(b_, d))
    ws = torch.empty((ws_bytes + 3) // 4, device=dg.device, dtype=torch.float32) if ws_bytes else None
    with _timed("dkt_bn_param_grads_f32"):
        st = lib.dkt_bn_param_grads_f32(_p(dg), _p(db), _p(og), _p(ob), b_, d, _p(ws), ws_bytes, _stream())
    _lib.check(st, "dkt_bn_param_grads_f32")
    return og, ob


def affine_normalize(x: torch.Tensor, a: torch.Tensor, s: torch.Tensor):
    """Zn = y / max(|y|_2, 1e-12), y = a x + s, row by row; a, s: [D] or [B,D].  Returns (Zn [B,N,D], rnorm [B,N]) (dkt_affine_normalize_f32: the front end of
    episodes with more than 128 rows, whose Gram kernels take unit rows as input)."""
    x = _req(x, "x", 3)
    b_, n, d = x.shape
    a = _req(a, "a")
    s = _req(s, "s")
    stride = _ab_stride(a, s, b_, d)
    zn = torch.empty_like(x)
    rnorm = torch.empty((b_, n), device=x.device, dtype=torch.float32)
    lib = _lib_now()
    with _timed("dkt_affine_normalize_f32"):
        st = lib.dkt_affine_normalize_f32(_p(x), _p(a), _p(s), stride, _p(zn), _p(rnorm), b_, n, d, _stream())
    _lib.check(st, "dkt_affine_normalize_f32")
    return zn, rnorm


def normalize_bn_bwd(dzn, zn, x, a, rnorm, mean=None, rstd=None):
    """Backward of affine_normalize (+ the train-mode batch statistics behind a, s when mean / rstd are given): (dX, dgamma_part [B,D], dbeta_part [B,D]) --
    the parts are None without statistics (dkt_normalize_bn_bwd_f32)."""
    dzn = _req(dzn, "dzn", 3)
    zn = _req(zn, "zn", 3)
    x = _req(x, "x", 3)
    b_, n, d = x.shape
    a = _req(a, "a")
    rnorm = _req(rnorm, "rnorm", 2)
    stride = _ab_stride(a, a, b_, d)
    train = mean is not None
    if train:
        mean = _req(mean, "mean", 2)
        rstd = _req(rstd, "rstd", 2)
    dx = torch.empty_like(x)
    dg = torch.empty((b_, d), device=x.device, dtype=torch.float32) if train else None
    db = torch.empty((b_, d), device=x.device, dtype=torch.float32) if train else None
    ws = torch.empty((b_, n), device=x.device, dtype=torch.float32)
    lib = _lib_now()
    with _timed("dkt_normalize_bn_bwd_f32"):
        st = lib.dkt_normalize_bn_bwd_f32(_p(dzn), _p(zn), _p(x), _p(a), stride, _p(mean), _p(rstd), _p(rnorm), _p(dx), _p(dg), _p(db), _p(ws), b_, n, d, _stream())
    _lib.check(st, "dkt_normalize_bn_bwd_f32")
    return dx, dg, db




class _EpisodeLossBnFn(torch.autograd.Function):
    """Training episode straight from the trunk output X: [BatchNorm1d(train) +] F.normalize + linear Gram
    (dkt_gram_bn_train_f32; DKT_FUSED_STATS=0: dkt_bn_stats_f32 + dkt_gram_bn_f32) -> MLL (dkt_mll_f32) ; backward dkt_gram_bn_bwd_f32.  The normalised features are
    never written to memory.  use_bn=False is the plain cossim kernel (no bn_out: affine map = identity).
    More than 128 rows (the 20-way shapes): dkt_bn_stats_f32 -> dkt_affine_normalize_f32 (Zn written once: the large-N Gram kernels take unit rows as input) ->
    dkt_gram_f32 -> dkt_mll_f32; backward dkt_gram_bwd_f32 -> dkt_normalize_bn_bwd_f32."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, use_bn, y, sv, mean, noise, cls_weight, jitter0, max_tries):
        b_, n, d = x.shape
        ctx.lowrank = lowrank_applies(n, d, y.shape[-2], b_, front_end=True)
        ctx.big = n > FUSED_EP_MAX_N or ctx.lowrank
        if ctx.big:
            if use_bn:
                st = bn_stats(x, gamma, beta, eps)
                a, s, bmean, rstd, bvar = st["a"], st["s"], st["mean"], st["rstd"], st["var_unbiased"]
            else:
                a = torch.ones(d, device=x.device, dtype=torch.float32)
                s = torch.zeros(d, device=x.device, dtype=torch.float32)
                bmean = rstd = bvar = torch.zeros(0, device=x.device, dtype=torch.float32)
            zn, rnorm = affine_normalize(x, a, s)
            if ctx.lowrank:
                # D <= 64 < N (Conv4S / Omniglot): the episode in feature space on the normalised features -- no E, no W (csrc/dkt_lowrank.hip)
                cw_ = _req(cls_weight.reshape(-1), "cls_weight", 1)
                o = _lowrank_forward(zn, _req(y, "y"), _req(sv.reshape(-1), "sv", 1), _req(mean.reshape(-1), "mean", 1), _req(noise.reshape(-1), "noise", 1),
                                     cw_, jitter0, max_tries, True)
                ctx.use_bn = bool(use_bn)
                ctx.save_for_backward(x, zn, o["wd"], a, s, bmean, rstd, rnorm, o["dsv"], o["dmean"], o["dnoise"], cw_, o["v"], o["t"])
                ctx.shapes = (sv.shape, mean.shape, noise.shape, None if gamma is None else gamma.shape, None if beta is None else beta.shape)
                ctx.mark_non_differentiable(o["logp"], o["alpha"], o["info"], o["jitter"], bmean, bvar, a, s, rnorm)
                ctx.set_materialize_grads(False)
                return o["obj"], o["logp"], o["alpha"], o["info"], o["jitter"], None, bmean, bvar, a, s, rnorm
            e = gram(zn, None, KERNEL_LINEAR_UNIT)
            out = mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cls_weight, jitter0=jitter0, max_tries=max_tries)
            obj = objective(out["logp"], cls_weight)
            ctx.use_bn = bool(use_bn)
            ctx.save_for_backward(x, zn, out["w"], a, s, bmean, rstd, rnorm, out["dsv"], out["dmean"], out["dnoise"], cls_weight)
            ctx.shapes = (sv.shape, mean.shape, noise.shape, None if gamma is None else gamma.shape, None if beta is None else beta.shape)
            ctx.mark_non_differentiable(out["logp"], out["alpha"], out["info"], out["jitter"], e, bmean, bvar, a, s, rnorm)
            ctx.set_materialize_grads(False)
            return obj, out["logp"], out["alpha"], out["info"], out["jitter"], e, bmean, bvar, a, s, rnorm
        if use_bn and os.environ.get("DKT_FUSED_STATS", "1") != "0":
            e, rnorm, st = gram_bn_train(x, gamma, beta, eps)               # statistics + Gram in one pass over x
            a, s, bmean, rstd, bvar = st["a"], st["s"], st["mean"], st["rstd"], st["var_unbiased"]
        else:
            if use_bn:
                st = bn_stats(x, gamma, beta, eps)
                a, s, bmean, rstd, bvar = st["a"], st["s"], st["mean"], st["rstd"], st["var_unbiased"]
            else:
                a = torch.ones(d, device=x.device, dtype=torch.float32)
                s = torch.zeros(d, device=x.device, dtype=torch.float32)
                bmean = rstd = bvar = torch.zeros(0, device=x.device, dtype=torch.float32)
            e, rnorm = gram_bn(x, a, s)
        out = mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cls_weight, jitter0=jitter0, max_tries=max_tries)
        obj = objective(out["logp"], cls_weight)
        ctx.use_bn = bool(use_bn)
        ctx.save_for_backward(x, e, out["w"], a, s, bmean, rstd, rnorm, out["dsv"], out["dmean"], out["dnoise"], cls_weight)
        ctx.shapes = (sv.shape, mean.shape, noise.shape, None if gamma is None else gamma.shape, None if beta is None else beta.shape)
        ctx.mark_non_differentiable(out["logp"], out["alpha"], out["info"], out["jitter"], e, bmean, bvar, a, s, rnorm)
        ctx.set_materialize_grads(False)
        return obj, out["logp"], out["alpha"], out["info"], out["jitter"], e, bmean, bvar, a, s, rnorm

    @staticmethod
    def backward(ctx, gobj, *_unused):
        if gobj is None:
            return (None,) * 12
        x, e, w, a, s, bmean, rstd, rnorm, dsv, dmean, dnoise, cw = ctx.saved_tensors[:12]
        gobj = gobj.contiguous()
        if ctx.big:                                           # (the second saved tensor is Zn here; feature-space path: the third is W', then V and T)
            dzn = _lowrank_backward(e, ctx.saved_tensors[12], ctx.saved_tensors[13], w, gobj) if ctx.lowrank else gram_bwd(w, e, gobj, unit_rows=True, w_symmetric=True)
            dx, dg, db = normalize_bn_bwd(dzn, e, x, a, rnorm, bmean if ctx.use_bn else None, rstd if ctx.use_bn else None)
        elif ctx.use_bn:
            dx, dg, db = gram_bn_bwd(w, e, x, a, s, rnorm, bmean, rstd, gobj)
        else:
            dx, dg, db = gram_bn_bwd(w, e, x, a, s, rnorm, None, None, gobj)
        ng = ctx.needs_input_grad
        ggamma = gbeta = None
        if dg is not None and ((ng[1] and ctx.shapes[3] is not None) or (ng[2] and ctx.shapes[4] is not None)):
            sg, sb = bn_param_grads(dg, db)
            ggamma = sg.reshape(ctx.shapes[3]) if (ng[1] and ctx.shapes[3] is not None) else None
            gbeta = sb.reshape(ctx.shapes[4]) if (ng[2] and ctx.shapes[4] is not None) else None
        gsv, gmean, gnoise = hyper_grads(gobj, cw, dsv if ng[6] else None, dmean if ng[7] else None, dnoise if ng[8] else None, ctx.shapes[:3])
        return (dx if ng[0] else None), ggamma, gbeta, None, None, None, gsv, gmean, gnoise, None, None, None


def episode_loss_bn(x, gamma, beta, y, sv, mean, noise, cls_weight, eps: float = 1e-5, jitter0: float = 1e-6, max_tries: int = 3,
                    use_bn: bool = True, full: bool = False):
    """x:[B,N,D] trunk output BEFORE bn_out.  Returns (obj[B], logp, alpha, info, jitter, E (None when the episode ran in feature space: lowrank_applies), batch_mean[B,D],
    batch_var_unbiased[B,D]) -- the last two feed the caller's running-statistics update -- plus, with full=True, the folded
    affine map a, s and the row scales rnorm (zn = (a x + s) rnorm: what a caller needs to re-create the normalised features)."""
    out = _EpisodeLossBnFn.apply(x, gamma, beta, eps, use_bn, y, sv, mean, noise, cls_weight, jitter0, max_tries)
    return out if full else out[:8]


def episode_loss_linear(z, y, sv, mean, noise, cls_weight, jitter0: float = 1e-6, max_tries: int = 3, unit_rows: bool = False):
    """z:[B,N,D] (already bn_out'ed + normalised).  Returns (obj[B], logp, alpha, info, jitter, E).
    unit_rows=True: z went through F.normalize (cossim / bncossim), |z| <= 1 element-wise -- the Gram kernels may then use
    the scaled 2-way f16 split (same fp32-level accuracy, less staging work).
    D <= 64 < N where it wins (lowrank_applies: the Conv4S / Omniglot episodes -- every 420-row batch, 105-row batches from 3072 episodes): the episode runs in
    feature space and E is None -- no N x N matrix exists."""
    if z.dim() == 3 and lowrank_applies(z.shape[1], z.shape[2], y.shape[-2], z.shape[0]):
        return _EpisodeLossLowRankFn.apply(z, y, sv, mean, noise, cls_weight, jitter0, max_tries, bool(unit_rows)) + (None,)
    return _EpisodeLossLinearFn.apply(z, y, sv, mean, noise, cls_weight, jitter0, max_tries, unit_rows)
