#!/bin/bash
# The evidence of a round in one GPU call (gpurun): rocprofv3 stats + PMC passes per config, the band path's timings / kernel breakdown / phase clocks, the GPU suite,
# the graded bench line, the small-batch probes, train_loop per episode.  Everything lands under gpurun_out/; copy what is to be judged into profiles/r0N/.
cd $GRAFT_REPO_ROOT
bash tools/prof_all.sh cfg4 cfg4_n320 cfg2 cfg1 cfg1_20way cfg0 cfg3 > gpurun_out/prof_all.log 2>&1
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
# the B = 1 step as round 5 dispatched it (generic Gram kernels below 64 episodes, tensor-expression reductions): the "before" of docs/MEASUREMENTS.md R6b
mkdir -p gpurun_out/prof_b1_before; DKT_TWINS=1 DKT_GRAM_FEWEP=0 DKT_GRAM_EP_MINB=64 DKT_FUSED_REDUCTIONS=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b1_before -o b1 -- python tools/b1_step_trace.py 1 > gpurun_out/prof_b1_before/run.log 2>&1
