#!/bin/bash
# A/B of kernel-variant libraries on ONE box: tools/ab_lib.sh libA.so libB.so ...   (paths relative to the repo root)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib (rep $rep)"
    DKT_AMD_LIB=$ROOT/$lib python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], {k:v['ms'] for k,v in d['kernels'].items()})"
    DKT_AMD_LIB=$ROOT/$lib python tools/mll_occupancy.py 2>/dev/null | grep -E "B=  256 grad=1|B= 1024 grad=1|B= 2048 grad=1"
  done
done
