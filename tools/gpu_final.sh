#!/bin/bash
# round 2: full GPU suite + smoke + the bench line (modes, parity, cpu baseline)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log; tail -14 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; echo "bench exit $?"
python tools/bench_brief.py gpurun_out/r2_bench.json bench
