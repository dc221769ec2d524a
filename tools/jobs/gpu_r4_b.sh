#!/bin/bash
# round 4, GPU call B: tests after the split-K hand-off fix, plan sweep, VLAD / k-means stages, staging rates
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_vit.py tests/test_gpu_vlad_topk.py tests/test_gpu_vlad_cache.py tests/test_gpu_x6.py -m gpu -q --durations=5 < /dev/null > gpurun_out/r4b_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r4b_pytest.log; tail -30 gpurun_out/r4b_pytest.log | cut -c1-250
timeout 600 python tools/sweep_b1.py 1,2 < /dev/null > gpurun_out/r4b_b1_plan_sweep.log 2> gpurun_out/r4b_b1_plan_sweep.err
grep -E "round-3|default plans|BEST" gpurun_out/r4b_b1_plan_sweep.log | cut -c1-400; tail -3 gpurun_out/r4b_b1_plan_sweep.err
timeout 300 python tools/run_stage.py vlad_61img vlad_256img kmeans_5Mx1536 --check < /dev/null > gpurun_out/r4b_stages.log 2>&1; cut -c1-600 gpurun_out/r4b_stages.log
timeout 300 python tools/time_staging.py < /dev/null > gpurun_out/r4b_staging.log 2>&1; cat gpurun_out/r4b_staging.log
