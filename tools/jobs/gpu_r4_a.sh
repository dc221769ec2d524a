#!/bin/bash
# round 4, GPU call A: full GPU suite on the split-K / staging / telemetry build, the small-M plan sweep, a bench run
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 < /dev/null > gpurun_out/r4a_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r4a_pytest.log; tail -25 gpurun_out/r4a_pytest.log | cut -c1-300
timeout 600 python tools/sweep_b1.py 1,2,4 < /dev/null > gpurun_out/r4a_b1_plan_sweep.log 2> gpurun_out/r4a_b1_plan_sweep.err
grep -E "round-3|default plans|BEST" gpurun_out/r4a_b1_plan_sweep.log | cut -c1-400; tail -3 gpurun_out/r4a_b1_plan_sweep.err
timeout 900 python bench.py --steps 6 --warmup 2 --no-modes < /dev/null > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench exit $?"
python tools/bench_brief.py gpurun_out/r4a_bench.json bench | cut -c1-900
tail -5 gpurun_out/r4a_bench.err
