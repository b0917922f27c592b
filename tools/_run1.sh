cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest_gpu_all.log 2>&1; tail -8 gpurun_out/r3a/pytest_gpu_all.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3a/bench.log 2>&1; tail -1 gpurun_out/r3a/bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'ms_per_step', 'valid', 'deterministic')}, d.get('mll_rel_err'))
for k, v in d['kernels'].items(): print(k, v)
"
