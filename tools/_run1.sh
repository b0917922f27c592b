cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest_gpu_all2.log 2>&1; tail -6 gpurun_out/r3a/pytest_gpu_all2.log
timeout 600 python bench.py --config cfg0 --no-other-configs --no-cpu-baseline --no-test-time 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'ms_per_step', 'valid', 'deterministic')})
for k, v in d['kernels'].items(): print(' ', k, v, d['roofline_by_kernel'][k]['frac'])
"
