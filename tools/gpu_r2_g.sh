#!/bin/bash
# round 2: attention 64-query waves A/B, GEMM 256x256 config end to end
mkdir -p gpurun_out
export TMPDIR=/tmp
for a in 0 1; do
  ANYLOC_ATTN_H3_CFG=$a timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention_h3" 2>&1 | tail -1
  ANYLOC_ATTN_H3_CFG=$a timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes > gpurun_out/r2_bench_attn$a.json 2> gpurun_out/r2_bench_attn$a.err
  python tools/bench_brief.py gpurun_out/r2_bench_attn$a.json attn_cfg=$a
done
ANYLOC_H3_CFG=2 timeout 600 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_x6.py -m gpu -q -x -k "h3" 2>&1 | tail -2
ANYLOC_H3_CFG=2 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes > gpurun_out/r2_bench_h3cfg2.json 2> gpurun_out/r2_bench_h3cfg2.err
python tools/bench_brief.py gpurun_out/r2_bench_h3cfg2.json h3cfg=2
