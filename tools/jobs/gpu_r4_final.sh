#!/bin/bash
# round 4 validation of a build (one gpurun call): full GPU suite, smoke(), the full bench line, rocprofv3 kernel stats of the
# bench command, and the PMC passes over the B = 61 forward (FETCH / WRITE / SQ counters -> per-kernel tables + pmc_traffic.json)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q --durations=6 < /dev/null > gpurun_out/r4_pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r4_pytest_gpu.log; tail -12 gpurun_out/r4_pytest_gpu.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/r4_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/r4_smoke.log; tail -2 gpurun_out/r4_smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 < /dev/null > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; echo "bench exit $?"
python tools/bench_brief.py gpurun_out/r4_bench.json bench | cut -c1-900
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes --no-stages < /dev/null > $R/gpurun_out/r4_prof_bench.json 2> $R/gpurun_out/r4_prof_bench.err
cd $R; f=$(find gpurun_out/r4_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4_kernel_stats.csv && head -8 gpurun_out/r4_kernel_stats.csv | cut -c1-200
python tools/bench_brief.py gpurun_out/r4_prof_bench.json profiled | head -1 | cut -c1-300
if [ -z "$SKIP_PMC" ]; then bash tools/gpu_pmc_vit.sh 2>&1 | tail -14 | cut -c1-300; fi
