#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vlad_topk.py -m gpu -q -x --durations=3 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
timeout 600 python tools/bench_configs.py 4 > gpurun_out/config4.log 2>&1; tail -2 gpurun_out/config4.log | cut -c1-900
