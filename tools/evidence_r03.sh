# Round-3 evidence run (GPU box, via gpurun): logs -> gpurun_out/r03/, copied to profiles/r03/ afterwards.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $O/v1_pytest_gpu.log
( timeout 900 python bench.py 2>&1 | grep "^{" ) > $O/v1_bench_full.json
( timeout 600 python tools/time_mll_h2.py 2>&1 | grep -v amdgpu.ids ) > $O/v1_time_mll_h2_vs_f32mfma.log
( DKT_MLL_H2E_MINB=1 timeout 600 python tools/time_mll_batch.py 2>&1 | grep -v amdgpu.ids; DKT_MLL_H2E_MINB=1000000000 timeout 600 python tools/time_mll_batch.py 2>&1 | grep -v amdgpu.ids ) > $O/v1_time_mll_batch.log
( for a in "1024 105" "8192 105" "8192 85"; do timeout 300 python tools/mll_h2e_clocks.py $a 2>&1 | grep -v amdgpu.ids; done ) > $O/v1_mll_h2e_phase_clocks.log
( timeout 300 python tools/mll_phase_clocks.py 8192 5 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/mll_phase_clocks.py 8192 5 f32mfma 2>&1 | grep -v amdgpu.ids ) > $O/v1_mll_wpm_phase_clocks.log
( timeout 300 python tools/ubench_h2.py 2>&1 | grep -v amdgpu.ids ) > $O/v1_ubench_f16_mfma.log
( timeout 300 python tools/time_nonlinear.py 2>&1 | grep -v amdgpu.ids ) > $O/v1_time_nonlinear_one_launch.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $O/v1_smoke.log
tail -3 $O/v1_pytest_gpu.log; cat $O/v1_ubench_f16_mfma.log; cat $O/v1_smoke.log
