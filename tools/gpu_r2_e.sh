#!/bin/bash
# round 2: new tests (cache paths, one-GPU distributed, PCA, ingest, K=200) + config3 bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_vlad_cache.py tests/test_gpu_distributed_one_gpu.py tests/test_gpu_pca.py tests/test_gpu_kernels.py tests/test_gpu_vlad_topk.py tests/test_gpu_vit.py -m gpu -q --durations=5 > gpurun_out/r2_newtests.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r2_newtests.log; grep -E "passed|failed|Error|error|assert|^E " gpurun_out/r2_newtests.log | cut -c1-260 | tail -40
timeout 900 python bench.py --workload config3 --steps 2 --warmup 1 > gpurun_out/r2_bench_config3.json 2> gpurun_out/r2_bench_config3.err; echo "config3 exit $?"
cut -c1-1500 gpurun_out/r2_bench_config3.json; tail -5 gpurun_out/r2_bench_config3.err | cut -c1-300
