#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 0 10 12 13 11; do
  ANYLOC_GEMM_CFG=$c timeout 300 python tools/microbench_gemm.py 32 2>&1 | tail -1
done | tee gpurun_out/gemm_sweep3.log
for c in 12 13; do ANYLOC_GEMM_CFG=$c timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k gemm 2>&1 | tail -2; done | tee -a gpurun_out/gemm_sweep3.log
ANYLOC_GEMM_CFG=12 timeout 300 python tools/microbench_gemm.py 61 2>&1 | tail -1 | tee -a gpurun_out/gemm_sweep3.log
