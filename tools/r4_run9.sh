#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "trunk or fused or front_end or 20way or meta_batch or drivers or abi" > gpurun_out/r9_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r9_pytest.log
tail -12 gpurun_out/r9_pytest.log
