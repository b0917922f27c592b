#!/bin/bash
# Runs on the GPU box (via gpurun): for every BASELINE config a rocprofv3 kernel-trace + stats run of bench.py's own step and separate
# PMC passes (FETCH_SIZE, WRITE_SIZE; for the headline config also the SQ / TCC counters); compact summaries ->
# gpurun_out/prof_r06/<config>_summary.txt (copied to profiles/r05/ and condensed into profiles/pmc_traffic.json by
# tools/make_traffic_json.py).   usage: tools/prof_all.sh [configs...]
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CONFIGS=${@:-cfg2 cfg1 cfg3 cfg0 cfg4 cfg4_n320}
for cfg in $CONFIGS; do
  ARGS="--config $cfg --no-other-configs --no-cpu-baseline --no-test-time"
  rm -rf $OUT/$cfg; mkdir -p $OUT/$cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$cfg/stats -- python $ROOT/bench.py $ARGS > $OUT/$cfg/stats.log 2>&1
  PMCS=("FETCH_SIZE" "WRITE_SIZE")
  if [ $cfg = cfg2 ]; then
    PMCS+=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE")
  fi
  for pmc in "${PMCS[@]}"; do
    tag=$(echo $pmc | tr ' ' '_' | cut -c1-40)
    timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/$cfg/pmc_$tag -- python $ROOT/bench.py $ARGS --steps 5 --warmup 2 > $OUT/$cfg/pmc_$tag.log 2>&1
  done
  python - $OUT $cfg <<'PY'
import csv, glob, os, collections, sys
out, cfg = sys.argv[1], sys.argv[2]
with open("%s/%s_summary.txt" % (out, cfg), "w") as f:
    f.write("# rocprofv3 summary of `python bench.py --config %s --no-other-configs --no-cpu-baseline --no-test-time` (tools/prof_all.sh)\n" % cfg)
    for p in sorted(glob.glob("%s/%s/stats/**/*kernel_stats.csv" % (out, cfg), recursive=True)):
        f.write("== kernel_stats.csv (kernel trace of the DEFAULT step counts)\n" + open(p).read() + "\n")
    bl = [l for l in open("%s/%s/stats.log" % (out, cfg)) if l.startswith("{")]
    if bl:
        f.write("== bench line of the traced run\n" + bl[-1] + "\n")
    for d in sorted(glob.glob("%s/%s/pmc_*" % (out, cfg))):
        if not os.path.isdir(d):
            continue
        for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(p)):
                k = (r.get("Kernel_Name", "?")[:90], r.get("Counter_Name", "?"))
                agg[k][0] += 1
                agg[k][1] += float(r.get("Counter_Value", 0) or 0)
            f.write("== counters (--pmc pass %s, --steps 5 --warmup 2)\n" % os.path.basename(d)[4:])
            for (kn, cn), (n, v) in sorted(agg.items()):
                if any(t in kn for t in ("dkt", "gram", "mll", "tiled", "band", "bgemm", "chol", "rbf", "sqdist", "big_", "lowrank", "objective_kernel", "hyper_grads")):
                    f.write("%-90s %-28s dispatches %4d  mean %.6g\n" % (kn, cn, n, v / max(n, 1)))
print(open("%s/%s_summary.txt" % (out, cfg)).read()[:3000])
PY
  rm -rf $OUT/$cfg
done
