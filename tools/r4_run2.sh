#!/bin/bash
# round 4, GPU run 2: parity suite with the resident-accumulator W kernel, its A/B against the block-column kernel, the full bench line
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4b
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/check_tiled.py > $OUT/check_tiled.log 2>&1
tail -25 $OUT/check_tiled.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for wres in 1 0; do
  rm -rf $OUT/st_$wres
  DKT_MLL_TILED_WRES=$wres timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$wres -- python $ROOT/tools/time_tiled.py > $OUT/st_$wres.log 2>&1
  python - $OUT/st_$wres <<'PY' > $OUT/stats_wres$wres.txt
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:10]:
    print("%-64s calls %5s total %9.1f us avg %9.1f us %5.1f%%" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
  cat $OUT/st_$wres.log | grep "B=" ; cat $OUT/stats_wres$wres.txt
  rm -rf $OUT/st_$wres
done
cd $ROOT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json
