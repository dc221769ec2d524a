#!/bin/bash
# round 4, GPU call L: chain-order MFMA issue in gemm_x6 (x6 mode bench A/B) and gemm_h3m (config-3 shard A/B)
mkdir -p gpurun_out
export TMPDIR=/tmp
cp anyloc_amd/libanyloc_hip.so /tmp/lib_orig.so
for rep in 1 2; do for v in chain x6chain; do
  cp tools/ab_libs/lib_$v.so anyloc_amd/libanyloc_hip.so
  timeout 300 python bench.py --gemm x6 --steps 6 --warmup 2 --no-modes --no-stages --no-cpu-baseline < /dev/null > gpurun_out/abx_${v}_$rep.json 2>> gpurun_out/abx.err
  python tools/bench_brief.py gpurun_out/abx_${v}_$rep.json "x6mode:$v#$rep" | head -1 | cut -c1-200
done; done
for rep in 1 2; do for v in x6chain hmchain; do
  cp tools/ab_libs/lib_$v.so anyloc_amd/libanyloc_hip.so
  timeout 300 python tools/run_stage.py config3_shard 2>&1 | grep config3 | cut -c1-160 | sed "s/^/$v#$rep /"
done; done
cp /tmp/lib_orig.so anyloc_amd/libanyloc_hip.so
