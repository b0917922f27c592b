#!/bin/bash
# per-class base matrices through the tile-array pipeline: new tests + the neighbours they could break
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "per_class or tile_array or class_kernel or every_kernel or bench_batch_cfg4 or large_episode" > gpurun_out/r7_pytest.log 2>&1
echo "pytest exit $?" >> gpurun_out/r7_pytest.log
tail -15 gpurun_out/r7_pytest.log
timeout 600 python tools/pc_time.py > gpurun_out/r7_pc_time.log 2>&1
tail -20 gpurun_out/r7_pc_time.log
