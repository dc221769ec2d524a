#!/bin/bash
# round 4, GPU call E: the two-fp16-plane few-query kernel (tests + timing against the bf16 one)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -k "few_queries or config2" < /dev/null > gpurun_out/r4e_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r4e_pytest.log; tail -25 gpurun_out/r4e_pytest.log | cut -c1-250
for rep in 1 2; do for o in 1 2; do
ANYLOC_OPTIONS=topk_fewq_x6=$o timeout 200 python tools/time_topk.py 2>&1 | grep "\"nq\": 61" | cut -c1-300; done; done
