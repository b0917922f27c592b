"""Per-kernel durations of the band marginal-likelihood path by call shape, from a rocprofv3 kernel trace of tools/check_band.py time (tools/prof_band.sh).
    python tools/band_trace_breakdown.py gpurun_out/prof_<tag>/<tag>_kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seqs = collections.defaultdict(lambda: collections.defaultdict(list))
cur = None
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "band_init" in n:
        cur = (r["Grid_Size_X"], r["Grid_Size_Y"])
    if cur and "band" in n:
        seqs[cur][n.split("::")[1].split("(")[0]].append(d)
    if "generic" in n:
        cur = None
for k in sorted(seqs, key=lambda k: (-int(k[1]), -int(k[0]))):
    tot = 0.0
    print("init grid %s x %s episodes" % k)
    for kn, v in seqs[k].items():
        v = sorted(v)
        print("   %-28s n=%3d  median %9.1f us  min %9.1f  max %9.1f" % (kn, len(v), v[len(v) // 2], v[0], v[-1]))
