#!/bin/bash
# round 4, GPU call D: full GPU suite + smoke + full bench on the build with the plan table, fused3 VLAD at every parts count, kmeans_update
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --durations=6 < /dev/null > gpurun_out/r4d_pytest.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r4d_pytest.log; tail -16 gpurun_out/r4d_pytest.log | cut -c1-250
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/r4d_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/r4d_smoke.log; tail -2 gpurun_out/r4d_smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 < /dev/null > gpurun_out/r4d_bench.json 2> gpurun_out/r4d_bench.err; echo "bench exit $?"
python tools/bench_brief.py gpurun_out/r4d_bench.json bench | cut -c1-1200
tail -3 gpurun_out/r4d_bench.err
