#!/bin/bash
# round 4, GPU run 3: W kernel v2 (packed tile table, strip groups): parity + per-kernel time
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r4c
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/check_tiled.py > $OUT/check_tiled.log 2>&1
grep -v amdgpu.ids $OUT/check_tiled.log | tail -14
timeout 600 python -m pytest tests -m gpu -x -q -k "tile_array or blocked_path or bench_batch or cfg4" > $OUT/pytest_sel.log 2>&1
tail -5 $OUT/pytest_sel.log
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/st
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -- python $ROOT/tools/time_tiled.py > $OUT/st.log 2>&1
python - $OUT/st <<'PY' > $OUT/stats.txt
import csv, glob, sys
p = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
import collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(p[0])):
    agg[(r["Kernel_Name"][:70], r.get("Grid_Size", r.get("Grid_Size_X", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, gsz), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print("%-70s grid %-9s calls %3d avg %9.1f us" % (k, gsz, len(v), sum(v) / len(v)))
PY
grep "B=" $OUT/st.log; cat $OUT/stats.txt
rm -rf $OUT/st
