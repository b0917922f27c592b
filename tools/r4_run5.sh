#!/bin/bash
# round 4: full GPU parity suite + smoke + the default bench line
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-r4h}
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -6 $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], {k: v["ms"] for k, v in d["kernels"].items()}, d["valid"], d.get("mll_rel_err"))
for k, v in d["other_configs"].items():
    print(k, v["value"], v["ms_per_step"], {a: b["ms"] for a, b in v["kernels"].items()}, v["valid"])
print(d.get("cpu_baseline", {}).get("by_threads"))
print(d.get("rccl_selftest"))
PY
