#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_kernels.py -m gpu -q --durations=3 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log | cut -c1-250
timeout 600 python tools/bench_configs.py 4 > gpurun_out/config4.log 2>&1; tail -1 gpurun_out/config4.log | cut -c1-900
timeout 300 python tools/microbench.py 2>&1 | grep -A8 "^vlad" | cut -c1-200
ANYLOC_VLAD_TWO_PASS=1 timeout 300 python tools/microbench.py 2>&1 | grep "^vlad" | cut -c1-200
