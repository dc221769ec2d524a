#!/bin/bash
mkdir -p gpurun_out
for c in 0 4 11 1; do
  ANYLOC_GEMM_CFG=$c timeout 300 python tools/microbench_gemm.py 61 2>&1 | tail -1
done | tee gpurun_out/gemm_sweep4.log
