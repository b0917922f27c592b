"""Build + ctypes binding of libdkt_hip.so (the C ABI declared in include/dkt_abi.h).

There is NO CPU fallback: every entry point of `ops` goes through this library, and `load()`
raises if the shared object is missing or a symbol of the header is absent.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(_ROOT, "include")
LIB_PATH = os.path.join(_HERE, "libdkt_hip.so")
SOURCES = ["dkt_gram.hip", "dkt_gram_ep.hip", "dkt_mll.hip", "dkt_mll_mfma.hip", "dkt_mll_reg.hip", "dkt_mll_big.hip", "dkt_predict.hip",
           "dkt_spectral.hip", "dkt_frontend.hip"]
# measurement-only kernels (stream ceilings, co-residency spinners): a separate test / tooling library, NOT part of the product
DIAG_SOURCES = ["dkt_diag.hip"]
DIAG_LIB_PATH = os.path.join(_HERE, "libdkt_diag.so")
HEADERS = [os.path.join(CSRC, "dkt_common.h"), os.path.join(CSRC, "dkt_mll.h"), os.path.join(CSRC, "dkt_tiles.h"),
           os.path.join(CSRC, "dkt_split.h"), os.path.join(INCLUDE, "dkt_abi.h")]
OBJ_DIR = os.path.join(_HERE, "build")

_c_p = ctypes.c_void_p
_c_i = ctypes.c_int
_c_f = ctypes.c_float

# name -> (restype, argtypes); must list EVERY function of include/dkt_abi.h (tests check this).
SIGNATURES = {
    "dkt_abi_version": (_c_i, []),
    "dkt_device_cu_count": (_c_i, []),
    "dkt_gram_f32": (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p]),
    "dkt_mll_workspace_bytes": (ctypes.c_size_t, [_c_i, _c_i, _c_i]),
    "dkt_mll_f32": (_c_i, [_c_p, _c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_f, _c_i,
                           ctypes.c_uint, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                           _c_p, ctypes.c_size_t, _c_p]),
    "dkt_gram_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p, ctypes.c_uint, _c_p]),
    "dkt_rbf_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_p]),
    "dkt_sqdist_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_p]),
    "dkt_predict_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "dkt_predict_var_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "dkt_bn_stats_f32": (_c_i, [_c_p, _c_p, _c_p, _c_f, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "dkt_gram_bn_f32": (_c_i, [_c_p, _c_p, _c_p, ctypes.c_long, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "dkt_gram_bn_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                                   _c_i, _c_i, _c_i, _c_p]),
    "dkt_smk_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "dkt_smk_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
}

_lock = threading.Lock()
_lib = None


def _lib_path() -> str:
    """DKT_AMD_LIB points the loader at an alternative build of the same ABI (A/B measurements of kernel variants)."""
    return os.environ.get("DKT_AMD_LIB") or LIB_PATH


def _flags() -> list:
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-I", INCLUDE, "-I", CSRC] + \
        os.environ.get("DKT_EXTRA_HIPCC_FLAGS", "").split()


def _digest(src: str) -> str:
    """Content hash of a source, every header and the flags: the object cache key (mtimes do not survive a checkout)."""
    import hashlib
    h = hashlib.sha256()
    for f in [src] + HEADERS:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(_flags()).encode())
    return h.hexdigest()[:20]


def _stamp(sources, replace=None) -> str:
    return ";".join(_digest((replace or {}).get(s, os.path.join(CSRC, s))) for s in sources)


def needs_build() -> bool:
    if os.environ.get("DKT_AMD_LIB"):
        return False
    if not os.path.exists(LIB_PATH) or not os.path.exists(LIB_PATH + ".stamp"):
        return True
    with open(LIB_PATH + ".stamp") as fh:
        return fh.read() != _stamp(SOURCES)


def _compile_link(sources, target, replace=None, verbose=False) -> str:
    """One hipcc -c per source (in parallel, objects cached by content hash under build/), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)

    def one(name):
        src = (replace or {}).get(name, os.path.join(CSRC, name))
        obj = os.path.join(OBJ_DIR, "%s.%s.o" % (os.path.basename(src), _digest(src)))
        if not os.path.exists(obj):
            cmd = [hipcc] + _flags() + ["-c", src, "-o", obj + ".tmp"]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s%s" % (name, res.stdout, res.stderr))
            os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(one, sources))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", target + ".tmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(target + ".tmp", target)
    with open(target + ".stamp", "w") as fh:
        fh.write(_stamp(sources, replace))
    return target


def build(force: bool = False, verbose: bool = False, out: str = None, replace: dict = None) -> str:
    """hipcc --offload-arch=gfx950 -> deep-kernel-transfer_amd/libdkt_hip.so (in-tree).
    Cross-compiles without a GPU.  `out` / `replace` ({source name: other path}) build a variant library for A/B runs."""
    if out is None and not force and not needs_build():
        return LIB_PATH
    if force and os.path.isdir(OBJ_DIR):
        for f in os.listdir(OBJ_DIR):
            if f.endswith(".o"):
                os.remove(os.path.join(OBJ_DIR, f))
    return _compile_link(SOURCES, out or LIB_PATH, replace, verbose)


def build_diag(verbose: bool = False) -> str:
    """The measurement-only kernels (tools/, tests): libdkt_diag.so, never loaded by the product path."""
    if os.path.exists(DIAG_LIB_PATH) and os.path.exists(DIAG_LIB_PATH + ".stamp"):
        with open(DIAG_LIB_PATH + ".stamp") as fh:
            if fh.read() == _stamp(DIAG_SOURCES):
                return DIAG_LIB_PATH
    return _compile_link(DIAG_SOURCES, DIAG_LIB_PATH, None, verbose)


def load() -> ctypes.CDLL:
    """dlopen the HIP library and bind every declared symbol; raises (never falls back) on failure."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = _lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                "libdkt_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "-- the DKT hot path has no CPU fallback." % path)
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise RuntimeError("libdkt_hip.so lacks symbol %s declared in include/dkt_abi.h" % name) from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def load_diag() -> ctypes.CDLL:
    """dlopen the measurement-only library (tools / tests); builds it on first use."""
    return ctypes.CDLL(build_diag())


STATUS = {0: "DKT_OK", -1: "DKT_ERR_BAD_ARG", -2: "DKT_ERR_TOO_LARGE", -3: "DKT_ERR_WORKSPACE", -4: "DKT_ERR_LAUNCH"}


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, STATUS.get(status, "?"), status))
