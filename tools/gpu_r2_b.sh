#!/bin/bash
# round 2, call 2: fused h3 pipeline bring-up + gemm_h3 tile sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" > gpurun_out/r2_attn.log 2>&1; tail -15 gpurun_out/r2_attn.log | cut -c1-250
timeout 300 python tools/debug_h3_fused.py dinov2_vitg14 2 4 322 2>&1 | tail -8
timeout 300 python tools/debug_h3_fused.py dinov2_vitl14 2 3 518 2>&1 | tail -8
timeout 300 python tools/debug_h3_fused.py dinov2_vits14 3 12 224 2>&1 | tail -8
for c in 0 2 3 4 5; do ANYLOC_H3_CFG=$c timeout 200 python tools/sweep_h3.py 2>&1 | tail -4; done | tee gpurun_out/r2_h3_sweep.log
for f in 0 1; do
  ANYLOC_H3_FUSE=$f timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_fuse$f.json 2> gpurun_out/r2_bench_fuse$f.err
  python tools/bench_brief.py gpurun_out/r2_bench_fuse$f.json fuse=$f
done
