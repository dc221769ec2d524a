#!/bin/bash
# interleaved A/B of whole library builds on the headline workload: bash tools/gpu_ab_libs.sh base chain ...   (tools/ab_libs/lib_<name>.so)
mkdir -p gpurun_out
export TMPDIR=/tmp
cp anyloc_amd/libanyloc_hip.so /tmp/lib_orig.so
for rep in $(seq 1 ${REPS:-2}); do
  for v in "$@"; do
    cp tools/ab_libs/lib_$v.so anyloc_amd/libanyloc_hip.so
    timeout 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-modes --no-stages --no-cpu-baseline < /dev/null > "gpurun_out/abl_${v}_$rep.json" 2>> gpurun_out/abl.err
    python tools/bench_brief.py "gpurun_out/abl_${v}_$rep.json" "$v#$rep" | head -1 | cut -c1-330
  done
done
cp /tmp/lib_orig.so anyloc_amd/libanyloc_hip.so
