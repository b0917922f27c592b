cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 600 python bench.py --config cfg1 --no-other-configs --no-cpu-baseline --no-test-time 2>&1 | grep "^{" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'ms_per_step', 'valid')}, {k: v['ms'] for k, v in d['kernels'].items()})
"; done
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $GRAFT_REPO_ROOT/bench.py --config cfg1 --no-other-configs --no-cpu-baseline --no-test-time > /tmp/p1.log 2>&1; find /tmp/p1 -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-200
