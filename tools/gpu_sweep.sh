#!/bin/bash
# gpurun: GEMM tile-configuration / ablation sweep.  usage: bash tools/gpu_sweep.sh "<cfgs>" "<batches>"
#   cfgs: ANYLOC_GEMM_CFG values (0 default; 1,2,3,4,11 alternative tiles; 5-8 timing-only ablations; 10 = 2-slab prefetch)
mkdir -p gpurun_out
for b in ${2:-32 61}; do
  for c in ${1:-0 4}; do
    ANYLOC_GEMM_CFG=$c timeout 300 python tools/microbench_gemm.py $b 2>&1 | tail -1
  done
done | tee gpurun_out/gemm_sweep.log
