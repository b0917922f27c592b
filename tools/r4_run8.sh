#!/bin/bash
# full GPU suite + smoke after the library split (mll_reg -> libdkt_diag.so, ABI 3)
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r8_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r8_pytest.log
tail -6 gpurun_out/r8_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
