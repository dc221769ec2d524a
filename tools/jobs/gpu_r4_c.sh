#!/bin/bash
# round 4, GPU call C: ring-depth sweep of the small-M plans, weight-residency probe, direct LayerNorm, VLAD kernel choice at 61 images
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_round4.py -m gpu -q -k "small or telemetry or transfers or call_pattern" < /dev/null > gpurun_out/r4c_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r4c_pytest.log; tail -12 gpurun_out/r4c_pytest.log | cut -c1-250
timeout 300 python tools/b1_mall_probe.py < /dev/null > gpurun_out/r4c_b1_mall_probe.log 2>&1; cut -c1-500 gpurun_out/r4c_b1_mall_probe.log
timeout 600 python tools/sweep_b1.py 1 < /dev/null > gpurun_out/r4c_b1_plan_sweep.log 2> gpurun_out/r4c_b1_plan_sweep.err
grep -E "round-3|default plans|BEST|others" gpurun_out/r4c_b1_plan_sweep.log | cut -c1-400; tail -3 gpurun_out/r4c_b1_plan_sweep.err
for o in "vlad_fused_v=0" "vlad_fused_v=4" "vlad_fused_v=4,vlad_parts=2" "vlad_fused_v=4,vlad_parts=8"; do
ANYLOC_OPTIONS=$o timeout 200 python tools/run_stage.py vlad_61img --check < /dev/null 2>&1 | grep vlad_61 | cut -c1-330; done
