#!/bin/bash
# round 3, GPU call E: few-query kernel (32-k and 64-k slabs) tests + timings; attention_h3 at 4 waves per SIMD A/B
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
for o in 1 2; do
ANYLOC_OPTIONS=topk_fewq_x6=$o timeout 600 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_distributed_one_gpu.py tests/test_gpu_round3.py -m gpu -q -k "topk or search or sharded or scheduling" < /dev/null > $O/e_pytest_fewq$o.log 2>&1; echo "exit: $?" >> $O/e_pytest_fewq$o.log; tail -4 $O/e_pytest_fewq$o.log | cut -c1-240
done
for opt in "topk_fewq_x6=0" "topk_fewq_x6=1" "topk_fewq_x6=2" "topk_fewq_x6=0" "topk_fewq_x6=1" "topk_fewq_x6=2"; do
  ANYLOC_OPTIONS=$opt timeout 300 python tools/time_topk.py 2>&1 | grep nq | head -1 | sed "s/^/$opt  /"
done | tee $O/e_fewq_ab.log
for rep in 1 2; do
  for opt in "attn_h3_occ=2" "attn_h3_occ=4"; do
    ANYLOC_OPTIONS=$opt timeout 300 python bench.py --steps 10 --warmup 2 --no-modes --no-stages --no-cpu-baseline < /dev/null > $O/e_ab_${opt}_$rep.json 2>> $O/e_ab.err
    python tools/bench_brief.py $O/e_ab_${opt}_$rep.json "$opt#$rep" | head -1 | cut -c1-420
  done
done
