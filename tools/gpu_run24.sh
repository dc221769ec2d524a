#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize_properties.py -m gpu -q 2>&1 | tail -3
timeout 600 python tools/microbench_batch.py 2>&1 | tail -7 | tee gpurun_out/batch2.log
for b in 32 16; do ANYLOC_GEMM_CFG=0 timeout 300 python tools/microbench_gemm.py $b 2>&1 | tail -1; done
