#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 0 5 6 7 8; do
  ANYLOC_GEMM_CFG=$c timeout 300 python tools/microbench_gemm.py 32 2>&1 | tail -1
done | tee gpurun_out/gemm_sweep2.log
for b in 30 61; do ANYLOC_GEMM_CFG=0 timeout 300 python tools/microbench_gemm.py $b 2>&1 | tail -1; done | tee -a gpurun_out/gemm_sweep2.log
