#!/bin/bash
# round 4, GPU run 4: PMC counters of the resident-accumulator W kernel (what binds it: LDS, issue, waits)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHAPE=1024,20,420,128
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_INSTS_WAVE32_LDS" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/p$i -- python $ROOT/tools/time_tiled.py $SHAPE > $OUT/p$i.log 2>&1
  tail -2 $OUT/p$i.log
done
python - <<PY | tee $OUT/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for p in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        kn = r.get("Kernel_Name", "?")
        if "tiled" not in kn:
            continue
        k = (kn[27:60], r.get("Counter_Name", "?"))
        agg[k][0] += 1
        agg[k][1] += float(r.get("Counter_Value", 0) or 0)
for (kn, cn), (n, v) in sorted(agg.items()):
    print("%-34s %-28s n %3d mean %.6g" % (kn, cn, n, v / max(n, 1)))
PY
rm -rf $OUT/p*/
