#!/bin/bash
# small-batch extractor: 64x64-tile GEMM threshold sweep (ANYLOC_H3_TINY_MAX = 128x128-tile count below which it is used)
mkdir -p gpurun_out
for t in 0 128 256 400 1000; do
  echo "## ANYLOC_H3_TINY_MAX=$t"
  ANYLOC_H3_TINY_MAX=$t timeout 300 python tools/microbench_batch.py 1,2,3,4,6 2>&1 | grep "B="
done | tee gpurun_out/r2_tiny_sweep.log
timeout 600 python -m pytest tests/test_gpu_fullsize_properties.py tests/test_gpu_x6.py -x -q -m gpu 2>&1 | tail -4
