#!/bin/bash
A="--no-other-configs --no-cpu-baseline --no-test-time --no-rccl-selftest"
for cfg in cfg3 cfg1; do
  for envs in "" "DKT_MLL_H2E_MINB=100000000" ""; do
    env $envs python bench.py --config $cfg $A 2>/dev/null | grep "^{" | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('$cfg', '[$envs]', j['value'], j['ms_per_step'], {k:round(v['ms'],4) for k,v in j['kernels'].items()})"
  done
done
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import dkt_amd
from dkt_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
for (b, c, n) in [(1024, 20, 320), (2048, 5, 150), (1024, 10, 250), (1024, 20, 420)]:
    per = n // c
    cls = torch.arange(c, device=dev).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
    sv = torch.full((c,), 0.7, device=dev); mean = torch.zeros(c, device=dev); noise = torch.full((c,), 0.1, device=dev)
    cw = torch.full((c,), -1.0 / (c * n), device=dev)
    z = torch.nn.functional.normalize(torch.randn(b, n, 128, generator=g, device=dev), dim=2).contiguous()
    e = ops.gram(z)
    for env in ({}, {"DKT_MLL_TILED_WGS": "2"}, {}):
        os.environ.update(env)
        for _ in range(2): ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
        torch.cuda.synchronize()
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3): ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
        t.record(); torch.cuda.synchronize()
        for k in env: os.environ.pop(k)
        print("mll B=%d C=%d N=%d %s: %.3f ms" % (b, c, n, env or "default (3 workgroups per CU)", s.elapsed_time(t) / 3), flush=True)
PY
