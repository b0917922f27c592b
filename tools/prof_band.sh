#!/bin/bash
# rocprofv3 kernel trace of the band marginal-likelihood path at the cfg4 shapes (tools/check_band.py time).  Usage: tools/prof_band.sh <tag>
cd /tmp && export TMPDIR=/tmp
tag=${1:-band}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python tools/check_band.py time > $out/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open("$out/summary.txt", "w") as fh:
    for r in rows[:25]:
        fh.write("%-110s calls %6s  total %10.3f ms  avg %9.3f us  %5s%%\n" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
print(open("$out/summary.txt").read())
PY
tail -12 $out/run.log
