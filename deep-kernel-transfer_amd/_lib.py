"""Build + ctypes binding of libdkt_hip.so (the C ABI declared in include/dkt_abi.h).

There is NO CPU fallback: every entry point of `ops` goes through this library, and `load()`
raises if the shared object is missing or a symbol of the header is absent.

Three in-tree shared objects, all hipcc --offload-arch=gfx950:
  libdkt_hip.so    the PRODUCT: the default kernel of every call, no measurement switch, no variant instantiation;
  libdkt_twins.so  the same sources with -DDKT_TWINS: every pipeline variant, legacy pipeline and validation twin the defaults were chosen from, selected
                   by the environment switches of DESIGN.md's appendix.  Same ABI.  Loaded by the tests / A-B tools only (DKT_TWINS=1 + a variant switch);
  libdkt_diag.so   measurement-only kernels (stream ceilings, co-residency spinners, the round-1 register-sweep kernel).
"""
from __future__ import annotations

import ctypes
import json
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(_ROOT, "include")
LIB_PATH = os.path.join(_HERE, "libdkt_hip.so")
SOURCES = ["dkt_gram.hip", "dkt_gram_ep.hip", "dkt_gram_big.hip", "dkt_gram_small.hip", "dkt_classkernel.hip", "dkt_mll.hip", "dkt_mll_mfma.hip", "dkt_mll_h2.hip", "dkt_mll_reg.hip", "dkt_mll_big.hip", "dkt_mll_tiled.hip", "dkt_mll_band.hip", "dkt_objective.hip", "dkt_predict.hip",
           "dkt_spectral.hip", "dkt_frontend.hip", "dkt_frontend_big.hip", "dkt_lowrank.hip"]
# measurement-only kernels (stream ceilings, co-residency spinners): a separate test / tooling library, NOT part of the product
DIAG_SOURCES = ["dkt_diag.hip", "dkt_mll_reg_twin.hip"]
DIAG_LIB_PATH = os.path.join(_HERE, "libdkt_diag.so")
TWINS_LIB_PATH = os.path.join(_HERE, "libdkt_twins.so")
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))) + [os.path.join(INCLUDE, "dkt_abi.h")]
OBJ_DIR = os.path.join(_HERE, "build")

_c_p = ctypes.c_void_p
_c_i = ctypes.c_int
_c_f = ctypes.c_float

# name -> (restype, argtypes); must list EVERY function of include/dkt_abi.h (tests check this).
SIGNATURES = {
    "dkt_abi_version": (_c_i, []),
    "dkt_device_cu_count": (_c_i, []),
    "dkt_reload_env": (None, []),
    "dkt_gram_f32": (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p]),
    "dkt_mll_workspace_bytes": (ctypes.c_size_t, [_c_i, _c_i, _c_i]),
    "dkt_mll_workspace_bytes_for": (ctypes.c_size_t, [_c_i, _c_i, _c_i, ctypes.c_uint]),
    "dkt_mll_f32": (_c_i, [_c_p, _c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_f, _c_i,
                           ctypes.c_uint, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                           _c_p, ctypes.c_size_t, _c_p]),
    "dkt_objective_f32": (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_p]),
    "dkt_hyper_grads_f32": (_c_i, [_c_p] * 8 + [_c_i, _c_i, _c_p]),
    "dkt_bn_param_grads_workspace_bytes": (ctypes.c_size_t, [_c_i, _c_i]),
    "dkt_bn_param_grads_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_p, ctypes.c_size_t, _c_p]),
    "dkt_gram_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p, ctypes.c_uint, _c_p]),
    "dkt_rbf_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_p]),
    "dkt_sqdist_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_p]),
    "dkt_predict_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "dkt_predict_per_class_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "dkt_predict_var_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "dkt_bn_stats_f32": (_c_i, [_c_p, _c_p, _c_p, _c_f, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "dkt_gram_bn_f32": (_c_i, [_c_p, _c_p, _c_p, ctypes.c_long, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "dkt_gram_bn_train_f32": (_c_i, [_c_p, _c_p, _c_p, _c_f, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "dkt_gram_bn_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                                   _c_i, _c_i, _c_i, _c_p]),
    "dkt_affine_normalize_f32": (_c_i, [_c_p, _c_p, _c_p, ctypes.c_long, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "dkt_normalize_bn_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "dkt_class_kernel_f32": (_c_i, [_c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "dkt_class_kernel_bwd_nsplit": (_c_i, [_c_i, _c_i]),
    "dkt_class_kernel_bwd_f32": (_c_i, [_c_p, _c_p, _c_i, _c_p, _c_i, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "dkt_lowrank_supported": (_c_i, [_c_i, _c_i, _c_i]),
    "dkt_lowrank_gram_f32": (_c_i, [_c_p, _c_p, ctypes.c_long, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "dkt_lowrank_noise_floor_f32": (_c_i, [_c_p, _c_p, _c_p, _c_f, _c_i, _c_p, _c_p, _c_i, _c_p]),
    "dkt_lowrank_finish_f32": (_c_i, [_c_p, _c_p, ctypes.c_long] + [_c_p] * 17 + [_c_i, _c_i, _c_i, _c_i, _c_p]),
    "dkt_lowrank_bwd_f32": (_c_i, [_c_p] * 6 + [_c_i, _c_i, _c_i, _c_i, _c_p]),
    "dkt_smk_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "dkt_smk_bwd_f32": (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
}

_lock = threading.Lock()
_libs = {}          # path -> bound CDLL


def _lib_path() -> str:
    """DKT_AMD_LIB points the loader at an alternative build of the same ABI (A/B measurements of kernel variants)."""
    return os.environ.get("DKT_AMD_LIB") or LIB_PATH


def _flags(twins: bool = False) -> list:
    return (["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-Rpass-analysis=kernel-resource-usage",
             "-I", INCLUDE, "-I", CSRC] + (["-DDKT_TWINS"] if twins else []) + os.environ.get("DKT_EXTRA_HIPCC_FLAGS", "").split())


# Register-spill budget per kernel of the PRODUCT library (regex on the mangled name -> max VGPR spills); everything else must not spill at all.  The check
# fails the build when a change makes the compiler spill inside a hot loop.  (The twins library carries the non-default instantiations -- NT = 8 MFMA twins
# with up to 188 spilled VGPRs among them -- and is not checked.)
SPILL_BUDGET = {
    # mll_h2e_kernel (wave per episode, the bench kernel since round 3): no entry = 0 spills allowed, at 254 of 256 VGPRs
    r"mll_h2_kernelILi7ELb1ELb1": 20,              # wave per matrix <NT = 7, GRAD, 5 waves per episode> at its 168-VGPR cap (batches < 1024 episodes): 16
    r"mll_h2_kernelILi[67]E": 12,                  # its forward-only / other-class-count instantiations: 8
    # (gram_sym_ep_split_kernel<8, 2, 2, ...>, the unit-row Gram forward for 112 < N <= 128, had 24 here until round 5: the spills sat in the slab loop, every
    #  scratch reload is a vmcnt(0) that drains the prefetch -- built for two workgroups per CU it has none and runs 1.89 -> 1.36 ms per 8192 episodes of 128 x 1600)
    r"gram_sym_ep_split_kernelILi[78]ELi1ELi1": 8,  # the bf16-split default at NT = 7 / 8
    # tile-array factorisation / inverse at 3 workgroups per CU (168 VGPRs), MC = 7 (N >= 384): values parked in scratch around the diagonal-tile sweep, none in the K loop
    r"tiled_factor_kernelILi\dELb1ELi3E": 28,
    r"tiled_invert_kernelILi\dELb1ELb1ELi3E": 24,
    # band reduction (QR + fused pass at 256 VGPRs): lane constants parked at kernel entry; the forward kernel reloads one per QR column, both a few per panel --
    # none inside the tile loop
    r"band_class_kernel": 8,                      # (only a -DDKT_BAND_CLASS_WAVES=6 measurement build spills: 4; the default, 5 waves per SIMD, has none)
    r"band_sym_kernelILb0E": 32,
    r"band_sym_kernelILb1E": 16,
}


def _parse_resource_remarks(text: str) -> dict:
    """-Rpass-analysis=kernel-resource-usage remarks -> {kernel: {vgprs, agprs, sgprs, scratch, vgpr_spill, sgpr_spill, lds, occupancy}}"""
    import re
    out, cur = {}, None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
            "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def check_resources(usage: dict) -> list:
    """Kernels whose VGPR spill count exceeds their budget (SPILL_BUDGET; default 0)."""
    import re
    bad = []
    floor = int(os.environ.get("DKT_SPILL_BUDGET_FLOOR", "0"))      # experiments with variant builds only (DKT_EXTRA_HIPCC_FLAGS)
    for name, u in usage.items():
        budget = floor
        for pat, b in SPILL_BUDGET.items():
            if re.search(pat, name):
                budget = max(b, floor)
                break
        if u.get("vgpr_spill", 0) > budget:
            bad.append((name, u.get("vgpr_spill", 0), budget))
    return bad


def _digest(src: str, twins: bool = False) -> str:
    """Content hash of a source, every header and the flags: the object cache key (mtimes do not survive a checkout)."""
    import hashlib
    h = hashlib.sha256()
    for f in [src] + HEADERS:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(_flags(twins)).encode())
    return h.hexdigest()[:20]


def _stamp(sources, replace=None, twins: bool = False) -> str:
    return ";".join(_digest((replace or {}).get(s, os.path.join(CSRC, s)), twins) for s in sources)


def needs_build() -> bool:
    if os.environ.get("DKT_AMD_LIB"):
        return False
    if not os.path.exists(LIB_PATH) or not os.path.exists(LIB_PATH + ".stamp"):
        return True
    with open(LIB_PATH + ".stamp") as fh:
        return fh.read() != _stamp(SOURCES)


def _compile_link(sources, target, replace=None, verbose=False, twins: bool = False, check: bool = True) -> str:
    """One hipcc -c per source (in parallel, objects cached by content hash under build/), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)

    def one(name):
        src = (replace or {}).get(name, os.path.join(CSRC, name))
        obj = os.path.join(OBJ_DIR, "%s.%s.o" % (os.path.basename(src), _digest(src, twins)))
        if not os.path.exists(obj) or not os.path.exists(obj + ".res.json"):
            cmd = [hipcc] + _flags(twins) + ["-c", src, "-o", obj + ".tmp"]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s%s" % (name, res.stdout, res.stderr))
            with open(obj + ".res.json", "w") as fh:
                json.dump(_parse_resource_remarks(res.stderr), fh)
            os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(one, sources))
    usage = {}
    for o in objs:
        with open(o + ".res.json") as fh:
            usage.update(json.load(fh))
    with open(os.path.join(OBJ_DIR, os.path.basename(target) + ".resource_usage.json"), "w") as fh:
        json.dump(usage, fh, indent=1, sort_keys=True)
    bad = check_resources(usage) if (check and not twins) else []       # the spill budget is the product library's
    if bad:
        raise RuntimeError("register spills beyond the budget (deep-kernel-transfer_amd/_lib.py SPILL_BUDGET):\n" +
                           "\n".join("  %s: %d VGPR spills (budget %d)" % b for b in bad))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", target + ".tmp"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(target + ".tmp", target)
    with open(target + ".stamp", "w") as fh:
        fh.write(_stamp(sources, replace, twins))
    if replace is None:
        # superseded objects of these sources (a header edit re-hashes every file: the cache would otherwise grow by ~ 20 MB per edit); kept: what the three
        # in-tree libraries were last linked from (this link's objects + whatever the other libraries' lists name)
        keep_path = os.path.join(OBJ_DIR, os.path.basename(target) + ".objects.json")
        with open(keep_path, "w") as fh:
            json.dump([os.path.basename(o) for o in objs], fh)
        keep = set()
        for f in os.listdir(OBJ_DIR):
            if f.endswith(".objects.json"):
                with open(os.path.join(OBJ_DIR, f)) as fh:
                    keep.update(json.load(fh))
        for f in os.listdir(OBJ_DIR):
            base = f[:-len(".res.json")] if f.endswith(".res.json") else f
            if base.endswith(".o") and base not in keep and any(base.startswith(src + ".") for src in SOURCES + DIAG_SOURCES):
                try:
                    os.remove(os.path.join(OBJ_DIR, f))
                except OSError:
                    pass
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


def build_twins(verbose: bool = False) -> str:
    """The same sources with -DDKT_TWINS (every variant / legacy pipeline / validation twin + the environment switches that select them): libdkt_twins.so,
    loaded by the tests and the A/B tools only."""
    if os.path.exists(TWINS_LIB_PATH) and os.path.exists(TWINS_LIB_PATH + ".stamp"):
        with open(TWINS_LIB_PATH + ".stamp") as fh:
            if fh.read() == _stamp(SOURCES, None, True):
                return TWINS_LIB_PATH
    return _compile_link(SOURCES, TWINS_LIB_PATH, None, verbose, twins=True)


def build_diag(verbose: bool = False) -> str:
    """The measurement-only kernels (tools/, tests): libdkt_diag.so, never loaded by the product path."""
    if os.path.exists(DIAG_LIB_PATH) and os.path.exists(DIAG_LIB_PATH + ".stamp"):
        with open(DIAG_LIB_PATH + ".stamp") as fh:
            if fh.read() == _stamp(DIAG_SOURCES):
                return DIAG_LIB_PATH
    return _compile_link(DIAG_SOURCES, DIAG_LIB_PATH, None, verbose, check=False)


def abi_version_of_header() -> int:
    """DKT_ABI_VERSION as include/dkt_abi.h declares it."""
    import re
    with open(os.path.join(INCLUDE, "dkt_abi.h")) as fh:
        return int(re.search(r"#define\s+DKT_ABI_VERSION\s+(\d+)", fh.read()).group(1))


def device_code_objects(path: str = None) -> list:
    """The gfx950 code objects (ELF images) inside a built library or object file: the uncompressed clang offload bundles of its .hip_fatbin data."""
    import struct
    blob = open(path or LIB_PATH, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], blob.find(magic)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(magic))[0]
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(magic, pos + len(magic))
    return out


_disasm_cache = {}


def device_disassembly(path: str = None) -> list:
    """llvm-objdump -d of every gfx950 code object of a library: a list of texts (cached per path + mtime; the audits below share it)."""
    import tempfile
    path = path or LIB_PATH
    key = (path, os.path.getmtime(path))
    if key not in _disasm_cache:
        objdump = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
        texts = []
        for img in device_code_objects(path):
            with tempfile.NamedTemporaryFile(suffix=".co") as fh:
                fh.write(img)
                fh.flush()
                res = subprocess.run([objdump, "-d", "--no-show-raw-insn", fh.name], capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError("llvm-objdump failed: " + res.stderr[:500])
            texts.append(res.stdout)
        _disasm_cache.clear()
        _disasm_cache[key] = texts
    return _disasm_cache[key]


# what the two disassembly audits below managed to parse (positive controls for their tests: a toolchain that prints branch operands or `// ADDR:` comments
# differently would otherwise make them pass without checking anything)
AUDIT_STATS = {}


def spill_reloads_in_streaming_loops(path: str = None) -> dict:
    """{kernel: number of scratch reloads} for every kernel that reloads a spilled register INSIDE a loop that also issues global / buffer loads.  Scratch traffic
    shares the in-order memory counter, and a reload's wait is `vmcnt(0)`: such a reload drains whatever the loop prefetched, on every trip (round 5: the unit-row Gram
    forward at 112 < N <= 128 ran 1.89 instead of 1.36 ms per 8192 episodes for 24 spilled registers).  Loops = backward branches of the disassembly."""
    import re
    rx_addr = re.compile(r"//\s*([0-9A-Fa-f]+):")
    rx_br = re.compile(r"^\s*s_c?branch\w*\s+(\d+)")
    out = {}
    AUDIT_STATS["branches"] = AUDIT_STATS["backward_branches"] = AUDIT_STATS["kernels"] = 0
    for text in device_disassembly(path):
        kernel, insns = None, []

        def flush():
            if kernel is None or not insns:
                return
            addrs = [a for a, _ in insns]
            worst = 0
            for a, t in insns:
                m = rx_br.match(t)
                if not m:
                    continue
                AUDIT_STATS["branches"] += 1
                off = int(m.group(1))
                off = off - 65536 if off >= 32768 else off
                tgt = a + 4 + 4 * off
                if tgt >= a:
                    continue
                AUDIT_STATS["backward_branches"] += 1
                body = [tt for aa, tt in insns if tgt <= aa <= a]
                nsc = sum("scratch_load" in tt for tt in body)
                if nsc and any(("buffer_load" in tt) or ("global_load" in tt) for tt in body):
                    worst = max(worst, nsc)
            if worst:
                out[kernel] = worst

        for line in text.splitlines():
            if line.endswith(">:"):
                flush()
                kernel, insns = line.split("<")[-1][:-2], []
                AUDIT_STATS["kernels"] += 1
                continue
            m = rx_addr.search(line)
            if m and kernel is not None:
                insns.append((int(m.group(1), 16), line.split("//")[0]))
        flush()
    return out


def unprotected_wide_buffer_stores(path: str = None) -> list:
    """Disassemble the library's device code (llvm-objdump) and list every 12 / 16-byte BUFFER store with a REGISTER in its soffset field whose data
    registers are written by the next VALU instruction.  hipcc's hazard recogniser skips that form of the store (it inserts the wait state only for a
    literal soffset); on gfx950 the store then sends whatever the VALU wrote (round 5: dX of a software-pipelined fused backward came out as run-to-run
    garbage).  The kernels keep the scalar offset in the VGPR offset instead (bstore4, dkt_mfma_tiles.h); this is the audit that they all do."""
    import re
    rx_store = re.compile(r"^\s*buffer_store_dwordx[34]\s+([va])\[(\d+):(\d+)\],\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0|vcc_lo|vcc_hi)\b")
    rx_dst = re.compile(r"^\s*(v_\w+)\s+(([va])\[(\d+):(\d+)\]|([va])(\d+))\b")
    hits = []
    AUDIT_STATS["wide_stores_any"] = AUDIT_STATS["wide_stores_parsed"] = 0
    rx_any = re.compile(r"^\s*buffer_store_dwordx[34]\s+[va]\[(\d+):(\d+)\]")          # (data in VGPRs or in accumulation registers)
    for text in device_disassembly(path):
        kernel, pending = "?", None
        for line in text.splitlines():
            if line.endswith(">:"):
                kernel, pending = line.split("<")[-1][:-2], None
                continue
            text_ = line.split("//")[0]
            if not text_.strip():
                continue
            if pending is not None:
                m = rx_dst.match(text_)
                if m and not m.group(1).startswith(("v_cmp", "v_mfma", "v_readlane", "v_readfirstlane")):
                    cls = m.group(3) or m.group(6)
                    lo, hi = (int(m.group(4)), int(m.group(5))) if m.group(3) else (int(m.group(7)), int(m.group(7)))
                    if cls == pending[3] and lo <= pending[1] and hi >= pending[0]:
                        hits.append((kernel, pending[2].strip(), text_.strip()))
                pending = None
            m = rx_store.match(text_)
            AUDIT_STATS["wide_stores_any"] += ("buffer_store_dwordx3" in text_) or ("buffer_store_dwordx4" in text_)
            AUDIT_STATS["wide_stores_parsed"] += rx_any.match(text_) is not None          # (the operand syntax the audit's regular expressions expect)
            if m:
                pending = (int(m.group(2)), int(m.group(3)), text_, m.group(1))
    return hits


def load(path: str = None) -> ctypes.CDLL:
    """dlopen the HIP library (the product unless `path` / DKT_AMD_LIB says otherwise) and bind every declared symbol; raises (never falls back) on failure."""
    path = path or _lib_path()
    with _lock:
        lib = _libs.get(path)
        if lib is not None:
            return lib
        if not os.path.exists(path):
            raise RuntimeError(
                "%s is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "-- the DKT hot path has no CPU fallback." % path)
        lib = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise RuntimeError("%s lacks symbol %s declared in include/dkt_abi.h" % (os.path.basename(path), name)) from e
            fn.restype = res
            fn.argtypes = args
        want = abi_version_of_header()
        got = int(lib.dkt_abi_version())
        if got != want:
            raise RuntimeError("%s implements DKT_ABI_VERSION %d, include/dkt_abi.h declares %d: rebuild (python -c 'import __graft_entry__ as g; g.build()')"
                               % (path, got, want))
        _libs[path] = lib
        return lib


_twins_checked = False


def load_twins() -> ctypes.CDLL:
    """The variant / twin build of the same ABI (tests, A/B tools); built on first use where the sources are present (the staleness check -- a content hash of
    every source -- runs once per process, not per call)."""
    global _twins_checked
    if not _twins_checked:
        build_twins()
        _twins_checked = True
    return load(TWINS_LIB_PATH)


def load_diag() -> ctypes.CDLL:
    """dlopen the measurement-only library (tools / tests); builds it on first use."""
    return ctypes.CDLL(build_diag())


STATUS = {0: "DKT_OK", -1: "DKT_ERR_BAD_ARG", -2: "DKT_ERR_TOO_LARGE", -3: "DKT_ERR_WORKSPACE", -4: "DKT_ERR_LAUNCH"}


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, STATUS.get(status, "?"), status))
