#!/bin/bash
# Run on the GPU box via gpurun: full GPU test suite + micro-benchmarks, logs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>&1 | head -8 > gpurun_out/smi.log
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 600 python tools/microbench.py > gpurun_out/microbench.log 2>&1
echo "microbench exit: $?" >> gpurun_out/microbench.log
cat gpurun_out/microbench.log
