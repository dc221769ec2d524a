#!/bin/bash
# GPU validation of a build (one gpurun call): the test files named on the command line (default: all), smoke(), the
# full bench line (modes, stages, parity, cpu baseline) and the rocprofv3 kernel stats of the bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TESTS=${@:-tests}
timeout ${PYTEST_TIMEOUT:-2400} python -m pytest $TESTS -m gpu -q --durations=8 < /dev/null > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log; tail -14 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
if [ -z "$SKIP_BENCH" ]; then
timeout 900 python bench.py --steps ${STEPS:-20} --warmup 3 < /dev/null > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench exit $?"
python tools/bench_brief.py gpurun_out/final_bench.json bench | cut -c1-700
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes --no-stages < /dev/null > $R/gpurun_out/final_prof_bench.json 2> $R/gpurun_out/final_prof_bench.err
cd $R; f=$(find gpurun_out/final_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/final_kernel_stats.csv && head -9 gpurun_out/final_kernel_stats.csv | cut -c1-200
python tools/bench_brief.py gpurun_out/final_prof_bench.json profiled | head -1 | cut -c1-400
fi
