cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
( time timeout 900 python bench.py ) > gpurun_out/r3a/bench_full.log 2>&1; tail -4 gpurun_out/r3a/bench_full.log | cut -c1-300; grep "^{" gpurun_out/r3a/bench_full.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'ms_per_step', 'valid', 'deterministic')}, d.get('mll_rel_err'), d.get('cpu_baseline',{}).get('value'))
for k, v in d['kernels'].items(): print(' ', k, v, d['roofline_by_kernel'][k]['frac'])
for c, o in d['other_configs'].items():
    print(c, o['value'], o['ms_per_step'], o['valid'])
    for k, v in o['kernels'].items(): print('   ', k, v, o['roofline_by_kernel'][k]['frac'])
print(d['collective'])
"
