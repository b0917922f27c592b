"""Parse a tools/prof.sh summary.txt into pmc_traffic.json: HBM bytes per launch per ABI kernel, from the
rocprofv3 FETCH_SIZE / WRITE_SIZE passes.  Correction per MI355X_MICROARCH.md (HBM section): on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read -> x2 for the streaming
Gram kernels (validated: the result equals the algorithmic Z bytes to 0.02 %); WRITE_SIZE is used as is
(validated: equals the algorithmic dZ bytes).  The MFMA marginal-likelihood kernel stages E with the same 16-byte loads, so the same factor is
applied to it (an upper bound: its 64-byte row segments may be tallied at their true size)."""
import json
import re
import sys

src = sys.argv[1]
episodes = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
out = sys.argv[3] if len(sys.argv) > 3 else "pmc_traffic.json"
names = {"gram_sym_ep_split_kernel": "dkt_gram_f32", "gram_sym_ep_bf16x3_kernel": "dkt_gram_f32", "gram_bwd_ep_f16x2_kernel": "dkt_gram_bwd_f32", "gram_bwd_ep_bf16x3_kernel": "dkt_gram_bwd_f32", "gram_sym_ep_kernel": "dkt_gram_f32", "gram_nt_kernel": "dkt_gram_f32", "gram_bwd_ep_kernel": "dkt_gram_bwd_f32",
         "gram_bwd_kernel": "dkt_gram_bwd_f32", "mll_reg_kernel": "dkt_mll_f32", "mll_generic_kernel": "dkt_mll_f32",
         "mll_mfma_kernel": "dkt_mll_f32"}
vals = {}
for line in open(src):
    m = re.search(r"::(\w+)<.*?(FETCH_SIZE|WRITE_SIZE)\s+dispatches\s+\d+\s+mean\s+([0-9.e+]+)", line)
    if m and m.group(1) in names:
        vals.setdefault(names[m.group(1)], {})[m.group(2)] = float(m.group(3))
res = {"episodes_per_launch": episodes, "source": src, "unit": "bytes per launch",
       "correction": "FETCH_SIZE KB x 1024 x 2 (gfx950 wide-read under-count, 16-byte-per-lane loads) + WRITE_SIZE KB x 1024", "config": "cfg2", "kernels": {}}
for k, v in vals.items():
    f = v.get("FETCH_SIZE", 0.0) * 1024.0
    w = v.get("WRITE_SIZE", 0.0) * 1024.0
    corr = 2.0       # every kernel of the step reads with 16-byte-per-lane loads (the MFMA marginal-likelihood kernel included)
    res["kernels"][k] = {"fetch_bytes": f * corr, "write_bytes": w, "hbm_bytes": f * corr + w}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
