#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "tile_array or cfg4 or per_class or 20way" 2>&1 | tail -3
A="--no-other-configs --no-cpu-baseline --no-test-time --no-rccl-selftest"
for cfg in cfg4 cfg4_n320; do
  for envs in "" "DKT_MLL_TILED_WNW=4" ""; do
    env $envs python bench.py --config $cfg $A 2>/dev/null | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$cfg', '[$envs]', j['value'], j['ms_per_step'], {k:round(v['ms'],4) for k,v in j['kernels'].items()}, j.get('deterministic'))"
  done
done
