#!/bin/bash
# round 4, GPU run 1: memory-side-cache probe, chunk-size sweep of the tile-array MLL, its traffic-only (no-math) build, baseline bench
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4a
mkdir -p $OUT
cd $ROOT
python tools/mall_probe.py > $OUT/mall_probe.log 2>&1
for ch in 1024 256 128 64 32 26; do
  echo "== DKT_MLL_TILED_CHUNK=$ch" >> $OUT/chunk_sweep.log
  DKT_MLL_TILED_CHUNK=$ch timeout 200 python tools/time_tiled.py >> $OUT/chunk_sweep.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
for lib in libdkt_hip.so libdkt_nomath.so; do
  rm -rf $OUT/st_$lib
  DKT_AMD_LIB=$ROOT/deep-kernel-transfer_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$lib -- python $ROOT/tools/time_tiled.py > $OUT/st_$lib.log 2>&1
  python - $OUT/st_$lib <<'PY' > $OUT/stats_$lib.txt
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:12]:
    print("%-64s calls %5s total %9.1f us avg %9.1f us %5.1f%%" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
  rm -rf $OUT/st_$lib
done
cd $ROOT
timeout 600 python bench.py > $OUT/bench_baseline.json 2> $OUT/bench_baseline.err
tail -c 600 $OUT/bench_baseline.json
cat $OUT/mall_probe.log $OUT/chunk_sweep.log $OUT/stats_*.txt
