#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate passes) for one kernel: tools/pmc_traffic.sh mll|gram|gram_bwd
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-mll}
OUT=$ROOT/gpurun_out/traffic_$K
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pmc in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_32B_sum"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-30)
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/$tag -- python $ROOT/tools/run_one_kernel.py $K > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for p in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        kn = r.get("Kernel_Name", "?")
        if not any(t in kn for t in ("mll", "gram")):
            continue
        k = (kn[:60], r.get("Counter_Name", "?"))
        agg[k][0] += 1
        agg[k][1] += float(r.get("Counter_Value", 0) or 0)
for (kn, cn), (n, v) in sorted(agg.items()):
    print("%-60s %-24s n %3d mean %.6g" % (kn, cn, n, v / max(n, 1)))
PY
