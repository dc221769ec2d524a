#!/bin/bash
# gpurun: re-run GPU tests, smoke, bench, and rocprofv3 kernel trace of the bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --steps 6 --warmup 2 > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?"; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.log
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); echo $f; head -25 "$f"
