#!/bin/bash
# round 2: new tests (cache paths, one-GPU distributed, PCA, ingest, K=200) + config3 bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_pca.py -m gpu -q > gpurun_out/r2_newtests.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r2_newtests.log; grep -E "passed|failed|Error|error|assert|^E " gpurun_out/r2_newtests.log | cut -c1-260 | tail -40
true
cut -c1-1500 gpurun_out/r2_bench_config3.json; tail -5 gpurun_out/r2_bench_config3.err | cut -c1-300
