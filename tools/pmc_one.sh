#!/bin/bash
# PMC passes (separate runs, kernel-trace only) over one kernel: tools/pmc_one.sh mll|gram|gram_bwd
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-mll}
shift
ARGS="$@"
OUT=$ROOT/gpurun_out/pmc_${K}${DKT_PMC_TAG:-}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_LEVEL_WAVES SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/p$i -- python $ROOT/tools/run_one_kernel.py $K $ARGS > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
out = "$OUT"
agg = collections.defaultdict(lambda: [0, 0.0])
for p in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        kn = r.get("Kernel_Name", "?")
        if not any(t in kn for t in ("mll", "gram")):
            continue
        k = (kn[:70], r.get("Counter_Name", "?"))
        agg[k][0] += 1
        agg[k][1] += float(r.get("Counter_Value", 0) or 0)
with open(out + "/summary.txt", "w") as f:
    for (kn, cn), (n, v) in sorted(agg.items()):
        f.write("%-70s %-28s dispatches %3d mean %.6g\n" % (kn, cn, n, v / max(n, 1)))
print(open(out + "/summary.txt").read())
PY
