#!/bin/bash
# in-step A/B of the episode-resident variants: default (2223 / 1222) vs (22232 / 211) vs (2223 / 211)
A="--no-other-configs --no-cpu-baseline --no-test-time --no-rccl-selftest"
for cfg in cfg2 cfg1 cfg3; do
  for envs in "" "DKT_GRAM_BWD_UNIT_VAR=211" "DKT_GRAM_UNIT_VAR=22232 DKT_GRAM_BWD_UNIT_VAR=211" ""; do
    env $envs python bench.py --config $cfg $A 2>/dev/null | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$cfg', '[$envs]', j['value'], j['ms_per_step'], {k:round(v['ms'],4) for k,v in j['kernels'].items()})"
  done
done
