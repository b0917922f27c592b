cd $GRAFT_REPO_ROOT
bash tools/prof_all.sh cfg4 cfg4_n320 cfg2 cfg1 > gpurun_out/prof_all.log 2>&1
timeout 400 python tools/check_band.py time > gpurun_out/band_final_check_band.log 2>&1
bash tools/prof_band.sh final > /dev/null 2>&1
timeout 300 python tools/band_phase_clocks.py > gpurun_out/band_final_phase_clocks.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest_gpu.log 2>&1; tail -2 gpurun_out/final_pytest_gpu.log
timeout 700 python bench.py > gpurun_out/final_bench_line.json 2> gpurun_out/final_bench.err; wc -c gpurun_out/final_bench_line.json
cp gpurun_out/bench_detail.json gpurun_out/final_bench_detail.json
bash tools/b1_minb_probe.sh > gpurun_out/b1_probe_final.log 2>&1
bash tools/prof_b1.sh > /dev/null 2>&1
python tools/b1_cpu_overhead.py 1 > gpurun_out/b1_cpu_overhead.log 2>&1
timeout 300 python tools/time_train_loop.py > gpurun_out/final_train_loop.log 2>&1; cat gpurun_out/final_train_loop.log
