#!/bin/bash
# round 5, call N: the PCA fit's float64 products on the library's own kernel (csrc/pca_f64.hip): tests, the C caller, timing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pca.py tests/test_c_abi_host.py -m gpu -q < /dev/null 2>&1 | grep -E '^E  |passed|failed' | head -20 | tee gpurun_out/r5n_pca.log
timeout 300 tests/c_abi/build/abi_host 2>&1 | grep -i "pca\|abi_host" | tee -a gpurun_out/r5n_pca.log
timeout 600 python tools/time_pca.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5n_pca.log
