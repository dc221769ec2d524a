#!/bin/bash
# round 5, call F: the shifted accumulation with its 7-bit centre table in LDS and the remainder folded in every 8 tiles (no vector memory in the gather):
# VLAD tests, tokens-per-image sweep (slope / intercept), shift vs gather, the VLAD stage of the bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_vlad_cache.py tests/test_gpu_property.py tests/test_gpu_round4.py tests/test_c_abi_host.py -m gpu -q -x -k "vlad or VLAD or cpu_tensor or c_host" < /dev/null > gpurun_out/r5f_pytest_vlad.log 2>&1
echo "pytest(vlad) exit: $?" >> gpurun_out/r5f_pytest_vlad.log; tail -4 gpurun_out/r5f_pytest_vlad.log | cut -c1-250
timeout 300 python tools/probe_vlad_fixed.py < /dev/null > gpurun_out/r5f_vlad_fixed_cost.log 2>&1; cut -c1-200 gpurun_out/r5f_vlad_fixed_cost.log | tail -22
timeout 300 python tools/time_vlad_shift.py < /dev/null > gpurun_out/r5f_vlad_shift.log 2>&1; grep -E "random" gpurun_out/r5f_vlad_shift.log | cut -c1-175
timeout 300 python tools/run_stage.py vlad_61img vlad_256img --check < /dev/null > gpurun_out/r5f_stage_vlad.json 2>&1; cut -c1-600 gpurun_out/r5f_stage_vlad.json | tail -4
