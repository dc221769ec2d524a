#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "default threshold"; timeout 300 python tools/microbench_batch.py 2>&1 | grep "^B=" | tee gpurun_out/r2_batch_default.log
echo "h3 forced at every size"; ANYLOC_X6_MIN_ROWS=0 timeout 300 python tools/microbench_batch.py 2>&1 | grep "^B=" | head -4 | tee gpurun_out/r2_batch_h3forced.log
