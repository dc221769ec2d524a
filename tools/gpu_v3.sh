#!/bin/bash
# fused3 (fp16 screening + exact resolution): parity under the whole VLAD / k-means test files, then timing vs the defaults
mkdir -p gpurun_out
ANYLOC_KMEANS_FUSED_V=3 ANYLOC_VLAD_FUSED_V=3 timeout 900 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_vlad_cache.py tests/test_gpu_distributed_one_gpu.py -x -q -m gpu 2>&1 | tail -15
{
for v in 2 3; do ANYLOC_KMEANS_FUSED_V=$v timeout 300 python tools/time_kmeans.py all 2>&1 | grep '"rows"'; done
for v in 1 3; do ANYLOC_VLAD_FUSED_V=$v timeout 300 python tools/sweep_vlad.py 100000 2>&1 | grep '"vlad"' | sed "s/^/V=$v /"; done
} | tee gpurun_out/r2_v3.log
