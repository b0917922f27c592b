cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mll or jitter or golden or kernel_type or per_class" > gpurun_out/r3a/pytest_p2h.log 2>&1; tail -4 gpurun_out/r3a/pytest_p2h.log
DKT_MLL_P2_GUARD=-1 timeout 900 python tools/time_mll_h2.py 3 2>&1 | grep -v amdgpu.ids > gpurun_out/r3a/time_mll_p2h_grow.log; cat gpurun_out/r3a/time_mll_p2h_grow.log
