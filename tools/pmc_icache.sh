#!/bin/bash
# Instruction-cache / fetch counters of dkt_mll_f32 (separate PMC passes, kernel-trace only): tools/pmc_icache.sh [B]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/p$i -- python $ROOT/tools/run_one_kernel.py mll ${1:-8192} > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for p in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        kn = r.get("Kernel_Name", "?")
        if "mll" not in kn:
            continue
        k = (kn[:60], r.get("Counter_Name", "?"))
        agg[k][0] += 1
        agg[k][1] += float(r.get("Counter_Value", 0) or 0)
with open("$OUT/summary.txt", "w") as f:
    for (kn, cn), (n, v) in sorted(agg.items()):
        f.write("%-60s %-28s dispatches %3d mean %.6g\n" % (kn, cn, n, v / max(n, 1)))
print(open("$OUT/summary.txt").read())
PY
