#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log; tail -10 gpurun_out/pytest_gpu.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_b61.log 2> gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench_b61.log | cut -c1-3200
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3 -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
cd $R; f=$(find gpurun_out/prof3 -name "*kernel_stats.csv" | head -1); head -7 "$f" | cut -c1-220
