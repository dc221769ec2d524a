#!/bin/bash
# round 3, GPU call C: few-query retrieval with the on-the-fly bf16 split (tests + A/B), K-chunked fp16 panels, LN default
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_vlad_topk.py tests/test_gpu_vit.py tests/test_gpu_distributed_one_gpu.py -m gpu -q -x --durations=5 < /dev/null > $O/c_pytest.log 2>&1; echo "pytest exit: $?" >> $O/c_pytest.log; tail -12 $O/c_pytest.log | cut -c1-240
ANYLOC_OPTIONS=topk_fewq_x6=1 timeout 600 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_distributed_one_gpu.py -m gpu -q -k "topk or search or sharded" < /dev/null > $O/c_pytest_fewq.log 2>&1; echo "exit: $?" >> $O/c_pytest_fewq.log; tail -5 $O/c_pytest_fewq.log | cut -c1-240
for opt in "topk_fewq_x6=0" "topk_fewq_x6=1" "topk_fewq_x6=0" "topk_fewq_x6=1"; do
  ANYLOC_OPTIONS=$opt timeout 300 python tools/time_topk.py 2>&1 | tail -4 | sed "s/^/$opt  /"
done
ANYLOC_OPTIONS= timeout 400 python tools/run_stage.py config3_shard --check < /dev/null 2>> $O/c.err | cut -c1-900
