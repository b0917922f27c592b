#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_cfg4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/tools/time_cfg4.py > $OUT/stats.log 2>&1
python - <<PY
import csv, glob
p = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:14]:
    print("%-70s calls %5s total %9.1f us avg %8.1f us %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
