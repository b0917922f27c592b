"""Condense the per-config rocprofv3 summaries of tools/prof_all.sh (profiles/r03/<config>_summary.txt) into
profiles/pmc_traffic.json: HBM-side bytes per launch of every ABI function of the step, per BASELINE config, which bench.py reports
as `roofline.traffic`.

Counters: FETCH_SIZE / WRITE_SIZE (KB at the L2 <-> fabric boundary, separate --pmc passes).  Correction per MI355X_MICROARCH.md (HBM
section): on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read -> x 2 (validated on the Gram
kernels: the result equals the algorithmic Z bytes to 0.02 %); WRITE_SIZE is used as is (validated: equals the algorithmic dZ
bytes).  An ABI function that launches several kernels per call (the tile-array marginal likelihood: etile + factor + invert + w +
the fix-up launch) is the sum over its kernels, per call.

usage: python tools/make_traffic_json.py profiles/r03 [profiles/pmc_traffic.json]"""
import glob
import json
import os
import re
import sys

src_dir = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/pmc_traffic.json"


def abi_of(kernel: str):
    k = kernel
    if "rbf_bwd" in k:
        return "dkt_rbf_bwd_f32"
    if "sqdist_bwd" in k:
        return "dkt_sqdist_bwd_f32"
    if "gram_bwd" in k or "gram_bn_bwd" in k:
        return "dkt_gram_bwd_f32"
    if "gram_" in k:
        return "dkt_gram_f32"
    if any(t in k for t in ("mll_", "tiled_", "big_", "bgemm", "chol_inv")):
        return "dkt_mll_f32"
    return None


res = {"unit": "bytes per launch (ABI call)", "correction": "FETCH_SIZE KB x 1024 x 2 (gfx950 wide-read under-count, 16-byte-per-lane loads) + WRITE_SIZE KB x 1024",
       "configs": {}}
for path in sorted(glob.glob(os.path.join(src_dir, "*_summary.txt"))):
    cfg = os.path.basename(path)[:-len("_summary.txt")]
    episodes = None
    per_kernel = {}                                    # kernel -> {counter: (dispatches, mean)}
    for line in open(path):
        if line.startswith("{") and "episodes_per_step_per_gpu" in line:
            episodes = json.loads(line)["config"]["episodes_per_step_per_gpu"]
        m = re.match(r"^(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches\s+(\d+)\s+mean\s+([0-9.e+-]+)", line)
        if m:
            per_kernel.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
    if episodes is None or not per_kernel:
        continue
    fn = {}
    for kname, ctr in per_kernel.items():
        f = abi_of(kname)
        if f is None:
            continue
        d = fn.setdefault(f, {"kernels": {}, "calls": None})
        disp = max(v[0] for v in ctr.values())
        d["kernels"][kname] = {"dispatches": disp, "fetch_kb_mean": ctr.get("FETCH_SIZE", (0, 0.0))[1], "write_kb_mean": ctr.get("WRITE_SIZE", (0, 0.0))[1]}
    kernels = {}
    for f, d in fn.items():
        calls = min(k["dispatches"] for k in d["kernels"].values())
        fetch = sum(k["fetch_kb_mean"] * k["dispatches"] for k in d["kernels"].values()) / calls * 1024.0 * 2.0
        write = sum(k["write_kb_mean"] * k["dispatches"] for k in d["kernels"].values()) / calls * 1024.0
        kernels[f] = {"fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write, "calls_in_profile": calls,
                      "device_kernels": sorted(k.split("(")[0][:70] for k in d["kernels"])}
    res["configs"][cfg] = {"episodes_per_launch": episodes, "source": path, "kernels": kernels}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({c: {k: round(v["hbm_bytes"] / 1e9, 3) for k, v in d["kernels"].items()} for c, d in res["configs"].items()}, indent=1))
