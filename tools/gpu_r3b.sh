#!/bin/bash
# round 3, GPU call B: full GPU suite on the new defaults, A/B of attention_h3 K-batching and LayerNorm rows per wave, the
# bench line with the reworked stages
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=6 < /dev/null > $O/b_pytest.log 2>&1; echo "pytest exit: $?" >> $O/b_pytest.log; tail -12 $O/b_pytest.log | cut -c1-240
for rep in 1 2; do
  for opt in "attn_h3_kbatch=0" "attn_h3_kbatch=1" "ln_rows_per_wave=2" "ln_rows_per_wave=1" "attn_h3_kbatch=1,ln_rows_per_wave=2"; do
    ANYLOC_OPTIONS=$opt timeout 300 python bench.py --steps 10 --warmup 2 --no-modes --no-stages --no-cpu-baseline < /dev/null > $O/b_ab_${opt}_$rep.json 2>> $O/b_ab.err
    python tools/bench_brief.py $O/b_ab_${opt}_$rep.json "$opt#$rep" | cut -c1-420
  done
done
timeout 900 python bench.py --steps 20 --warmup 3 < /dev/null > $O/b_bench.json 2> $O/b_bench.err; echo "bench exit $?"; tail -3 $O/b_bench.err
python tools/bench_brief.py $O/b_bench.json bench
