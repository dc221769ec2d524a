#!/bin/bash
# small-batch kernels: bitwise A/B against the pre-change kernels, timing, and the tests that pin B=1 == batch
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for m in dinov2_vitg14 dinov2_vits14; do
echo "## defaults"; timeout 120 python tools/check_small_batch.py $m < /dev/null 2>&1 | grep sha
echo "## ANYLOC_H3_DEEP_MAX=0 ANYLOC_LN_SMALL_ROWS=0"; ANYLOC_H3_DEEP_MAX=0 ANYLOC_LN_SMALL_ROWS=0 timeout 120 python tools/check_small_batch.py $m < /dev/null 2>&1 | grep sha
done
echo "## per-kernel, defaults"; ANYLOC_BATCH_PROFILE=1 timeout 120 python tools/microbench_batch.py 1,2,4 < /dev/null 2>&1 | grep -v "^Seed\|amdgpu.ids" | cut -c1-300
} | tee gpurun_out/small_batch.log
timeout 300 python -m pytest tests/test_gpu_fullsize_properties.py tests/test_gpu_vit.py tests/test_gpu_kernels.py tests/test_gpu_x6.py -q -x -m gpu < /dev/null > gpurun_out/t3.log 2>&1; tail -3 gpurun_out/t3.log
