#!/bin/bash
# rocprofv3 kernel-trace statistics of the two other entries into the hot path (from trunk features; RBF with per-class lengthscales):
# gpurun_out/prof_r03/other_paths_<name>_stats.txt (copied to profiles/r03/).  Runs on the GPU box via gpurun.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for name in bench_frontend time_nonlinear; do
  rm -rf $OUT/op_$name
  if [ $name = bench_frontend ]; then ARGS="2048"; else ARGS=""; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/op_$name -- python $ROOT/tools/$name.py $ARGS > $OUT/op_$name.log 2>&1
  ( echo "# rocprofv3 --kernel-trace --stats -- python tools/$name.py $ARGS   (tools/prof_other_paths.sh)"; grep -v amdgpu.ids $OUT/op_$name.log | grep -v "^\[" | tail -12; echo "== kernel_stats.csv (top 14 by total time)"; for p in $(find $OUT/op_$name -name "*kernel_stats.csv"); do head -15 $p | cut -c1-260; done ) > $OUT/other_paths_${name}_stats.txt
  rm -rf $OUT/op_$name $OUT/op_$name.log
done
cat $OUT/other_paths_*_stats.txt | cut -c1-200 | head -60
