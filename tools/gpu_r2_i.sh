#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_fullsize_properties.py -m gpu -q -x -k "vlad or kmeans" 2>&1 | tail -2
ANYLOC_VLAD_FUSED=1 timeout 300 python tools/sweep_vlad.py 2>&1 | grep "^{" | tee gpurun_out/r2_vlad_sweep.log
ANYLOC_VLAD_TWO_PASS=1 timeout 300 python tools/sweep_vlad.py 2>&1 | grep "^{" | tee -a gpurun_out/r2_vlad_sweep.log
