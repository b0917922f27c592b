#!/bin/bash
# PMC passes over the tile-array marginal-likelihood kernels (tools/time_tiled.py): HBM traffic, MFMA / VALU busy, waits.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_tiled
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHAPE=${1:-1024,20,420,128}
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/p$i -- python $ROOT/tools/time_tiled.py $SHAPE > $OUT/p$i.log 2>&1
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
