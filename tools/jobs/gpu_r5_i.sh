#!/bin/bash
# round 5, call I: bisecting the run-to-run nondeterminism of the fused VLAD launch over library builds
mkdir -p gpurun_out
cp anyloc_amd/libanyloc_hip.so /tmp/lib_orig.so
for v in "$@"; do
  cp tools/ab_libs/lib_$v.so anyloc_amd/libanyloc_hip.so
  echo "=== $v" >> gpurun_out/r5i_stress.log
  timeout 300 python tools/stress_vlad.py 20 < /dev/null >> gpurun_out/r5i_stress.log 2>&1
done
cp /tmp/lib_orig.so anyloc_amd/libanyloc_hip.so
cut -c1-200 gpurun_out/r5i_stress.log | grep -v amdgpu.ids
