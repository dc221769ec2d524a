#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for a in 0 2; do
  ANYLOC_ATTN_H3_CFG=$a timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention_h3" 2>&1 | tail -1
  ANYLOC_ATTN_H3_CFG=$a timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes > gpurun_out/r2_bench_attnp$a.json 2> gpurun_out/r2_bench_attnp$a.err
  python tools/bench_brief.py gpurun_out/r2_bench_attnp$a.json attn_cfg=$a
done
