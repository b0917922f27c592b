#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the kernels behind the from-trunk and RBF paths:
# gpurun_out/prof_r05/other_paths_pmc.txt (copied to profiles/r0N/).  Runs on the GPU box via gpurun.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/other_paths_pmc.txt
for name in bench_frontend time_nonlinear; do
  if [ $name = bench_frontend ]; then ARGS="2048"; else ARGS=""; fi
  for pmc in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/pm_$name_$pmc
    timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pm_${name}_$pmc -- python $ROOT/tools/$name.py $ARGS > /dev/null 2>&1
    python - $OUT/pm_${name}_$pmc $name $pmc >> $OUT/other_paths_pmc.txt <<'PY'
import csv, glob, sys, collections
d, name, pmc = sys.argv[1:4]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r.get("Kernel_Name", "?")[:80]
        v = float(r.get("Counter_Value", 0) or 0)
        agg[k][0] += 1; agg[k][1] += v; agg[k][2] = max(agg[k][2], v)
print("== tools/%s.py, --pmc %s (KB per dispatch: mean over all dispatches of the tool's batch sizes / max = the largest batch)" % (name, pmc))
for k, (n, s, mx) in sorted(agg.items()):
    if any(t in k for t in ("gram_bn", "class_kernel", "mll_h2", "gram_bwd_ep", "gram_nt", "gram_sym")):
        print("%-80s dispatches %4d  mean %.6g  max %.6g" % (k, n, s / max(n, 1), mx))
PY
    rm -rf $OUT/pm_${name}_$pmc
  done
done
cat $OUT/other_paths_pmc.txt | cut -c1-170
