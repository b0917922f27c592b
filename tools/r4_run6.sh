#!/bin/bash
# round 4: the N > 128 Gram kernels (episode-resident forward, 128-row backward): parity + the cfg4 bench lines, A/B against the round-2 kernels
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-r4k}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "large_n or cfg4 or gram_bwd or gram_vs or full_size" > $OUT/pytest_sel.log 2>&1
tail -4 $OUT/pytest_sel.log
for cfg in cfg4 cfg4_n320; do
  for ab in "1 1" "0 0"; do
    set -- $ab
    DKT_GRAM_BIG_EP=$1 DKT_GRAM_BWD_ROWS8=$2 timeout 300 python bench.py --config $cfg --no-other-configs --no-cpu-baseline --no-test-time --no-rccl-selftest --steps 6 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$cfg big_ep=$1 rows8=$2', d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['kernels'].items()}, d['valid'])"
  done
done
