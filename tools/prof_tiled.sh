#!/bin/bash
# rocprofv3 kernel-trace split of the tile-array marginal-likelihood path (tools/time_tiled.py); summary -> gpurun_out/prof_tiled/summary.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_tiled
rm -rf $OUT/stats; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/tools/time_tiled.py "$@" > $OUT/stats.log 2>&1
tail -3 $OUT/stats.log
python - <<PY | tee $OUT/summary.txt
import csv, glob
p = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:12]:
    print("%-64s calls %5s total %9.1f us avg %9.1f us %5.1f%%" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
