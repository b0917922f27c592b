#!/bin/bash
mkdir -p gpurun_out
python tools/sweep_ep_variants.py 2>&1 | grep -v amdgpu.ids > gpurun_out/v14_ep_variant_sweep.log
timeout 900 python -m pytest tests -m gpu -x -q -k "gram" 2>&1 | tail -3
A="--no-other-configs --no-cpu-baseline --no-test-time --no-rccl-selftest"
for cfg in cfg2 cfg1 cfg3; do
  for envs in "" "DKT_GRAM_UNIT_VAR=2223 DKT_GRAM_BWD_UNIT_VAR=1222" ""; do
    env $envs python bench.py --config $cfg $A 2>/dev/null | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$cfg', '[$envs]', j['value'], j['ms_per_step'], {k:round(v['ms'],4) for k,v in j['kernels'].items()}, 'deterministic', j.get('deterministic'))" | tee -a gpurun_out/v14_ep_variant_sweep.log
  done
done
