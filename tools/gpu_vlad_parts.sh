#!/bin/bash
# fused VLAD with several workgroups per image: parity, then timing against the two-pass path
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_vlad_cache.py -x -q -m gpu 2>&1 | tail -15
for e in "X=0" "ANYLOC_VLAD_TWO_PASS=1" "ANYLOC_VLAD_PARTS=1" "ANYLOC_VLAD_PARTS=2" "ANYLOC_VLAD_PARTS=8"; do
  env $e timeout 300 python tools/sweep_vlad.py 500000 2>&1 | grep '"vlad"'
done | tee gpurun_out/r2_vlad_parts.log
