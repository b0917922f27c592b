#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for bench.py; summaries -> gpurun_out/prof
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-test-time ${BENCH_ARGS:-}"
# kernel trace of bench.py's DEFAULT step counts (the per-kernel averages are compared with the bench line's live HIP-event times)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $ROOT/bench.py --no-cpu-baseline --no-test-time ${BENCH_ARGS:-} > $OUT/stats.log 2>&1
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $pmc | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/pmc_$tag -- python $ROOT/bench.py $ARGS > $OUT/pmc_$tag.log 2>&1
done
rocprofv3 -L > $OUT/counters_list.txt 2>&1
# compact summaries
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof"
with open(out + "/summary.txt", "w") as f:
    for p in sorted(glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)):
        f.write("== " + os.path.relpath(p, out) + "\n" + open(p).read() + "\n")
    for d in sorted(glob.glob(out + "/pmc_*")):
        if not os.path.isdir(d):
            continue
        for p in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(p)):
                k = (r.get("Kernel_Name", "?")[:60], r.get("Counter_Name", "?"))
                agg[k][0] += 1
                agg[k][1] += float(r.get("Counter_Value", 0) or 0)
            f.write("== " + os.path.relpath(p, out) + "\n")
            for (kn, cn), (n, v) in sorted(agg.items()):
                if "dkt" in kn or "gram" in kn or "mll" in kn:
                    f.write("%-60s %-28s dispatches %4d  mean %.6g\n" % (kn, cn, n, v / max(n, 1)))
print(open(out + "/summary.txt").read()[:6000])
PY
