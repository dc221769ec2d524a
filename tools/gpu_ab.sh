#!/bin/bash
# Interleaved A/B of library options on the headline workload (one MI355X, via gpurun):
#   bash tools/gpu_ab.sh "h3_fast_silu=0" "h3_fast_silu=1" ...      (each value of ANYLOC_OPTIONS is run REPS times, interleaved)
# prints one bench_brief line per run (images/s, ms/step, per-kernel ms per step); JSON lines under gpurun_out/ab_*.json
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in $(seq 1 ${REPS:-2}); do
  for opt in "$@"; do
    ANYLOC_OPTIONS=$opt timeout 300 python bench.py --steps ${STEPS:-10} --warmup 2 --no-modes --no-stages --no-cpu-baseline < /dev/null > "gpurun_out/ab_${opt}_$rep.json" 2>> gpurun_out/ab.err
    python tools/bench_brief.py "gpurun_out/ab_${opt}_$rep.json" "$opt#$rep" | head -1 | cut -c1-420
  done
done
