#!/bin/bash
# round 6, validation of the final build (one gpurun call): full GPU suite, smoke(), the full bench line, rocprofv3 kernel stats of
# the bench command and of the configs[2] workload (screened retrieval)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${TAG:-r6z}
timeout 1800 python -m pytest tests -m gpu -q --durations=6 < /dev/null > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/${T}_pytest_gpu.log; tail -12 gpurun_out/${T}_pytest_gpu.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/${T}_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/${T}_smoke.log; tail -2 gpurun_out/${T}_smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench exit $?"
python tools/bench_brief.py gpurun_out/${T}_bench.json bench | cut -c1-700 | head -34
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes --no-stages --power-steps 0 < /dev/null > $R/gpurun_out/${T}_prof_bench.json 2> $R/gpurun_out/${T}_prof_bench.err
cd $R; f=$(find gpurun_out/${T}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${T}_kernel_stats.csv && head -8 gpurun_out/${T}_kernel_stats.csv | cut -c1-200
python tools/bench_brief.py gpurun_out/${T}_prof_bench.json profiled | head -1 | cut -c1-300
rm -rf gpurun_out/${T}_prof
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_prof3 -o c3 -- python $R/bench.py --workload config3 --steps 3 --warmup 1 < /dev/null > $R/gpurun_out/${T}_prof_config3.json 2> $R/gpurun_out/${T}_prof_config3.err
cd $R; f=$(find gpurun_out/${T}_prof3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${T}_config3_kernel_stats.csv && head -8 gpurun_out/${T}_config3_kernel_stats.csv | cut -c1-200
tail -c 1200 gpurun_out/${T}_prof_config3.json
rm -rf gpurun_out/${T}_prof3
