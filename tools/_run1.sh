cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "per_class or every_kernel or correct_with or kernel_type or drivers" > gpurun_out/r3a/pytest_epc.log 2>&1; tail -6 gpurun_out/r3a/pytest_epc.log
timeout 600 python tools/time_nonlinear.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3a/time_nonlinear.log; cat gpurun_out/r3a/time_nonlinear.log
