# Round-4 evidence run (GPU box, via gpurun): logs -> gpurun_out/r04/, copied to profiles/r04/ afterwards (tools/prof_all.sh writes the per-config summaries).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $O/v17_pytest_gpu.log
( timeout 900 python bench.py 2>/dev/null | grep "^{" ) > $O/v17_bench_full.json
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/v17_smoke.log
# (build/ does not travel to the GPU box: copy deep-kernel-transfer_amd/build/libdkt_hip.so.resource_usage.json to profiles/r04/resource_usage.json in the build container)
tail -3 $O/v17_pytest_gpu.log; cat $O/v17_smoke.log; tail -c 400 $O/v17_bench_full.json
