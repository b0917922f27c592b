#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of dkt_mll_f32 alone (B = 8192 cfg2): tools/pmc_traffic_mll.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/traffic_mll
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/$pmc -- python $ROOT/tools/run_one_kernel.py mll 8192 > $OUT/$pmc.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for p in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        kn = r.get("Kernel_Name", "?")
        if "mll" not in kn:
            continue
        k = (kn[:60], r.get("Counter_Name", "?"))
        agg[k][0] += 1
        agg[k][1] += float(r.get("Counter_Value", 0) or 0)
for (kn, cn), (n, v) in sorted(agg.items()):
    print("%-60s %-12s n %3d mean %.6g KB  (algorithmic E or W: 352800 KB)" % (kn, cn, n, v / max(n, 1)))
PY
