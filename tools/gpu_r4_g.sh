#!/bin/bash
# round 4, GPU call G: read-ahead k-loop of the small-M plans: bitwise test + sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_vit.py -m gpu -q -x -k "small" < /dev/null > gpurun_out/r4g_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r4g_pytest.log; tail -14 gpurun_out/r4g_pytest.log | cut -c1-250
timeout 900 python tools/sweep_b1.py 1,2 < /dev/null > gpurun_out/r4g_b1_plan_sweep.log 2> gpurun_out/r4g_b1_plan_sweep.err
grep -E "round-3|default plans|BEST" gpurun_out/r4g_b1_plan_sweep.log | cut -c1-420; tail -3 gpurun_out/r4g_b1_plan_sweep.err
