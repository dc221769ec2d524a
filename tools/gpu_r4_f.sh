#!/bin/bash
# round 4, GPU call F: consumer-side split-K (slabs reduced by the next LayerNorm): tests + sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_round4.py tests/test_gpu_fullsize_properties.py tests/test_gpu_property.py -m gpu -q -x < /dev/null > gpurun_out/r4f_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r4f_pytest.log; tail -14 gpurun_out/r4f_pytest.log | cut -c1-250
timeout 600 python tools/sweep_b1_consumer.py 1,2 < /dev/null > gpurun_out/r4f_b1_consumer_splitk.log 2> gpurun_out/r4f_b1_consumer_splitk.err
grep -E "in-GEMM|default plans|BEST" gpurun_out/r4f_b1_consumer_splitk.log | cut -c1-300; tail -3 gpurun_out/r4f_b1_consumer_splitk.err
