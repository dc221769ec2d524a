#!/bin/bash
# round 4, GPU call M: patch embedding on the two-term fp16 GEMM -- parity tests, B = 1 / 61 timing, headline A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round4.py -k patch_embedding tests/test_gpu_vit.py -x -q 2>&1 | tail -5
timeout 600 python tools/time_patch_embed.py > gpurun_out/patch_embed.log 2>&1; tail -12 gpurun_out/patch_embed.log
for rep in 1 2; do for v in 0 1; do
  ANYLOC_OPTIONS="h3_patch=$v" timeout 300 python bench.py --steps 10 --warmup 2 --no-modes --no-stages --no-cpu-baseline < /dev/null > gpurun_out/abp_${v}_$rep.json 2>> gpurun_out/abp.err
  python tools/bench_brief.py gpurun_out/abp_${v}_$rep.json "h3_patch=$v#$rep" | head -1 | cut -c1-220
done; done
