#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_fullsize_properties.py -m gpu -q -x -k "kmeans or vlad" 2>&1 | tail -2
timeout 300 python tools/stamp_kmeans.py 2>&1 | tail -8
for v in 2 1; do ANYLOC_KMEANS_FUSED_V=$v timeout 300 python tools/time_kmeans.py 2>&1 | grep "^{"; done
ANYLOC_VLAD_FUSED=1 timeout 300 python tools/sweep_vlad.py 100000 2>&1 | grep "^{" | head -5
