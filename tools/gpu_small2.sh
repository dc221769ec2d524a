#!/bin/bash
# thresholds of the deep-stage tiny GEMMs: 64x64 tiles below DEEP_MAX -> 4 k-blocks per stage, below DEEP2_MAX -> 2
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for cfg in "320 0" "320 700" "320 1300" "450 1300" "700 1300"; do set -- $cfg
echo "## ANYLOC_H3_DEEP_MAX=$1 ANYLOC_H3_DEEP2_MAX=$2"; ANYLOC_H3_DEEP_MAX=$1 ANYLOC_H3_DEEP2_MAX=$2 timeout 100 python tools/check_small_batch.py < /dev/null 2>&1 | grep sha
done
} | tee gpurun_out/small_batch2.log
