"""Condense the per-config rocprofv3 summaries of tools/prof_all.sh (profiles/r0N/<config>_summary.txt) into
profiles/pmc_traffic.json: HBM-side bytes per launch of every ABI function of the step, per BASELINE config, which bench.py reports
as `roofline.traffic`.

Counters: FETCH_SIZE / WRITE_SIZE (KB at the L2 <-> fabric boundary, separate --pmc passes).  Correction per MI355X_MICROARCH.md (HBM
section): on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read -> x 2 (validated on the Gram
kernels: the result equals the algorithmic Z bytes to 0.02 %); WRITE_SIZE is used as is (validated: equals the algorithmic dZ
bytes).  An ABI function that launches several kernels per call (the tile-array marginal likelihood: etile + factor + invert + w +
the fix-up launch) is the sum over its kernels, per call.

Device kernel -> ABI function is an EXPLICIT table (KERNEL_TABLE): a kernel of this library that the table does not know and that moved
more than 1 MB per dispatch aborts the run (round 3 matched by substrings and silently summed cfg0's `gram_small_bwd_kernel` into
dkt_gram_f32).  Kernels of other libraries (torch element-wise glue) are ignored.

usage: python tools/make_traffic_json.py profiles/r04,profiles/r05 [profiles/pmc_traffic.json]     (several directories: a later one overrides a config of an earlier one)"""
import glob
import json
import os
import re
import sys

src_dir = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/pmc_traffic.json"

# device kernel (name without namespace / template arguments / signature) -> ABI function of include/dkt_abi.h that launches it
KERNEL_TABLE = {
    # dkt_gram_f32
    "gram_sym_ep_split_kernel": "dkt_gram_f32", "gram_sym_ep_kernel": "dkt_gram_f32", "gram_sym_tiles_split_kernel": "dkt_gram_f32",
    "gram_sym_bigep_f16x2_kernel": "dkt_gram_f32", "gram_small_kernel": "dkt_gram_f32", "gram_nt_kernel": "dkt_gram_f32", "gram_sym_fewep_kernel": "dkt_gram_f32", "gram_bn_sym_ep_kernel": "dkt_gram_f32",
    "gram_bn_train_f16_kernel": "dkt_gram_bn_train_f32",
    # dkt_gram_bwd_f32
    "gram_bwd_ep_f16x2_kernel": "dkt_gram_bwd_f32", "gram_bwd_ep_bf16x3_kernel": "dkt_gram_bwd_f32", "gram_bwd_ep_kernel": "dkt_gram_bwd_f32",
    "gram_bwd_rows_f16x2_kernel": "dkt_gram_bwd_f32", "gram_bwd_big_ep_kernel": "dkt_gram_bwd_f32", "gram_small_bwd_kernel": "dkt_gram_bwd_f32",
    "gram_bwd_kernel": "dkt_gram_bwd_f32", "gram_bn_bwd_ep_kernel": "dkt_gram_bwd_f32",
    # dkt_mll_f32
    "mll_h2e_kernel": "dkt_mll_f32", "mll_h2_kernel": "dkt_mll_f32", "mll_mfma_kernel": "dkt_mll_f32", "mll_generic_kernel": "dkt_mll_f32", "mll_kappa_flag_kernel": "dkt_mll_f32",
    "tiled_etile_kernel": "dkt_mll_f32", "tiled_factor_kernel": "dkt_mll_f32", "tiled_invert_kernel": "dkt_mll_f32", "tiled_w_kernel": "dkt_mll_f32",
    "tiled_wres_kernel": "dkt_mll_f32", "tiled_invres_kernel": "dkt_mll_f32", "big_form_kernel": "dkt_mll_f32", "bgemm_kernel": "dkt_mll_f32", "chol_inv_block_kernel": "dkt_mll_f32",
    "big_trmv_kernel": "dkt_mll_f32", "big_finish_kernel": "dkt_mll_f32",
    "band_init_kernel": "dkt_mll_f32", "band_sym_kernel": "dkt_mll_f32", "band_class_kernel": "dkt_mll_f32", "band_chain_kernel": "dkt_mll_f32",
    "band_finish_kernel": "dkt_mll_f32", "band_reduce_kernel": "dkt_mll_f32",
    # the [B, C]-sized reductions around it (round 6)
    "objective_kernel": "dkt_objective_f32", "hyper_grads_kernel": "dkt_hyper_grads_f32",
    "bn_param_grads_kernel": "dkt_bn_param_grads_f32", "bn_param_grads_fold_kernel": "dkt_bn_param_grads_f32",
    # the element-wise chain rules
    "rbf_bwd_kernel": "dkt_rbf_bwd_f32", "sqdist_bwd_kernel": "dkt_sqdist_bwd_f32",
    "class_kernel_fwd": "dkt_class_kernel_f32", "class_kernel_bwd": "dkt_class_kernel_bwd_f32",
    "predict_kernel": "dkt_predict_f32", "predict_var_kernel": "dkt_predict_var_f32",
    # the feature-space episode (round 5)
    "lowrank_gram_kernel": "dkt_lowrank_gram_f32", "lowrank_finish_kernel": "dkt_lowrank_finish_f32", "lowrank_bwd_kernel": "dkt_lowrank_bwd_f32",
    "lowrank_noise_floor_kernel": "dkt_lowrank_noise_floor_f32",
}
OURS = re.compile(r"gram|mll_|tiled_|band_|bgemm|chol_inv|big_|rbf_bwd|sqdist|class_kernel|predict|smk_|bn_stats|lowrank|objective_kernel|hyper_grads|bn_param_grads")


def short_name(kernel: str) -> str:
    """`void (anonymous namespace)::gram_small_kernel<1, 2>(float const*, ...)` -> `gram_small_kernel<1, 2>`"""
    k = kernel.strip()
    if k.startswith("void "):
        k = k[5:]
    k = k.replace("(anonymous namespace)::", "")
    depth, end = 0, len(k)
    for idx, ch in enumerate(k):                       # cut the signature: the first '(' outside the template argument list
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            end = idx
            break
    return k[:end].strip()


def abi_of(kernel: str):
    return KERNEL_TABLE.get(short_name(kernel).split("<")[0])


def main():
    res = {"unit": "bytes per launch (ABI call)",
           "correction": "FETCH_SIZE KB x 1024 x 2 (gfx950 wide-read under-count, 16-byte-per-lane loads) + WRITE_SIZE KB x 1024", "configs": {}}
    for path in [p_ for d_ in src_dir.split(",") for p_ in sorted(glob.glob(os.path.join(d_, "*_summary.txt")))]:
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
            kb = sum(v[1] for v in ctr.values())
            if f is None:
                if OURS.search(kname) and kb > 1024.0:
                    raise SystemExit("make_traffic_json: %s: kernel `%s` (%.1f MB per dispatch) is not in KERNEL_TABLE" % (path, short_name(kname), kb / 1024.0))
                continue
            d = fn.setdefault(f, {"kernels": {}})
            disp = max(v[0] for v in ctr.values())
            d["kernels"][short_name(kname)] = {"dispatches": disp, "fetch_kb_mean": ctr.get("FETCH_SIZE", (0, 0.0))[1], "write_kb_mean": ctr.get("WRITE_SIZE", (0, 0.0))[1]}
        kernels = {}
        for f, d in fn.items():
            calls = min(k["dispatches"] for k in d["kernels"].values())      # ABI calls in the profile (every kernel of the function runs once per call and chunk)
            fetch = sum(k["fetch_kb_mean"] * k["dispatches"] for k in d["kernels"].values()) / calls * 1024.0 * 2.0
            write = sum(k["write_kb_mean"] * k["dispatches"] for k in d["kernels"].values()) / calls * 1024.0
            kernels[f] = {"fetch_bytes": fetch, "write_bytes": write, "hbm_bytes": fetch + write, "calls_in_profile": calls,
                          "device_kernels": {n_: {"dispatches": k["dispatches"], "fetch_bytes": k["fetch_kb_mean"] * 2048.0, "write_bytes": k["write_kb_mean"] * 1024.0}
                                             for n_, k in sorted(d["kernels"].items())}}
        res["configs"][cfg] = {"episodes_per_launch": episodes, "source": path, "kernels": kernels}
    # the fused front-end kernels of the from-trunk path (tools/prof_other_paths_pmc.sh -> other_paths_pmc.txt: tools/bench_frontend.py at 2048 cfg2 episodes; the
    # `max` column = the largest batch of the tool): config "cfg2_from_trunk"
    for d_ in src_dir.split(","):
        path = os.path.join(d_, "other_paths_pmc.txt")
        if not os.path.exists(path):
            continue
        sect, kb = None, {}
        for line in open(path):
            m = re.match(r"^== tools/(\w+)\.py, --pmc (FETCH_SIZE|WRITE_SIZE)", line)
            if m:
                sect = (m.group(1), m.group(2))
                continue
            m = re.match(r"^(.*?)\s+dispatches\s+\d+\s+mean\s+[0-9.e+-]+\s+max\s+([0-9.e+-]+)", line)
            if m and sect and sect[0] == "bench_frontend":
                for kn, fnm in (("gram_bn_train_f16_kernel", "dkt_gram_bn_train_f32"), ("gram_bn_bwd_ep_kernel", "dkt_gram_bn_bwd_f32")):
                    if kn in m.group(1):
                        kb.setdefault(fnm, {})[sect[1]] = float(m.group(2))
        if kb:
            res["configs"]["cfg2_from_trunk"] = {"episodes_per_launch": 2048, "source": path, "kernels": {
                f: {"fetch_bytes": v.get("FETCH_SIZE", 0.0) * 2048.0, "write_bytes": v.get("WRITE_SIZE", 0.0) * 1024.0,
                    "hbm_bytes": v.get("FETCH_SIZE", 0.0) * 2048.0 + v.get("WRITE_SIZE", 0.0) * 1024.0} for f, v in kb.items()}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({c: {k: round(v["hbm_bytes"] / 1e9, 3) for k, v in d["kernels"].items()} for c, d in res["configs"].items()}, indent=1))


if __name__ == "__main__":
    main()
