#!/bin/bash
# round 3, GPU call A: the refactored build -- full GPU suite, the new kernels behind their options (fp16 retrieval panels,
# fast SiLU epilogue) under the parity tests, the bench line with stages, and A/B timings of the two options
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 < /dev/null > $O/a_pytest.log 2>&1; echo "pytest exit: $?" >> $O/a_pytest.log; tail -15 $O/a_pytest.log | cut -c1-240
ANYLOC_OPTIONS=topk_h3=1 timeout 600 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_distributed_one_gpu.py tests/test_gpu_fullsize_parity.py -m gpu -q -k "topk or search or sharded or vitg_full_depth_vs_oracle" < /dev/null > $O/a_pytest_topk_h3.log 2>&1; echo "exit: $?" >> $O/a_pytest_topk_h3.log; tail -6 $O/a_pytest_topk_h3.log | cut -c1-240
ANYLOC_OPTIONS=h3_fast_silu=1 timeout 600 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_vit.py -m gpu -q < /dev/null > $O/a_pytest_fast_silu.log 2>&1; echo "exit: $?" >> $O/a_pytest_fast_silu.log; tail -6 $O/a_pytest_fast_silu.log | cut -c1-240
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $O/a_smoke.log 2>&1; echo "smoke exit $?" >> $O/a_smoke.log; tail -2 $O/a_smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 < /dev/null > $O/a_bench.json 2> $O/a_bench.err; echo "bench exit $?"; tail -3 $O/a_bench.err
python tools/bench_brief.py $O/a_bench.json bench
for rep in 1 2; do
  for opt in "h3_fast_silu=0" "h3_fast_silu=1"; do
    ANYLOC_OPTIONS=$opt timeout 300 python bench.py --steps 10 --warmup 2 --no-modes --no-stages --no-cpu-baseline < /dev/null > $O/a_ab_${opt}_$rep.json 2>> $O/a_ab.err
    python tools/bench_brief.py $O/a_ab_${opt}_$rep.json "$opt#$rep"
  done
done
for opt in "topk_h3=0" "topk_h3=-1"; do
  ANYLOC_OPTIONS=$opt timeout 400 python tools/run_stage.py config3_shard --check < /dev/null > $O/a_stage_c3_${opt}.json 2>> $O/a_ab.err
  cut -c1-700 $O/a_stage_c3_${opt}.json
done
