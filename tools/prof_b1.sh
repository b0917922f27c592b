cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_b1
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b1 -- python tools/b1_step_trace.py 1 > $out/run.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open("$out/summary.txt", "w") as fh:
    for r in rows[:40]:
        fh.write("%-100s calls %6s  total %10.3f ms  avg %9.3f us  %5s%%\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
print(open("$out/summary.txt").read())
PY
tail -5 $out/run.log
