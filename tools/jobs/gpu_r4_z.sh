#!/bin/bash
# round 4, last GPU call (about 10 box-minutes were left): the torch-free C caller of the ABI, the dispatch-timestamp
# trace of one-image forwards, then the full validation of HEAD (GPU suite, smoke, the whole bench line).  Most
# informative first: the call may be cut short by the budget.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
echo "== C caller of the ABI (no Python in the process)"
timeout 300 tests/c_abi/build/abi_host > gpurun_out/rz_abi_host.log 2>&1; echo "abi_host exit $?" | tee -a gpurun_out/rz_abi_host.log
cut -c1-230 gpurun_out/rz_abi_host.log
echo "== one-image forwards under rocprofv3 --kernel-trace"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rz_b1 -o b1 -- python $R/tools/b1_trace_target.py 12 < /dev/null > $R/gpurun_out/rz_b1_target.json 2> $R/gpurun_out/rz_b1_target.err
cd $R; cat gpurun_out/rz_b1_target.json
f=$(find gpurun_out/rz_b1 -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python tools/b1_gaps.py "$f" 12 > gpurun_out/rz_b1_gaps.md 2> gpurun_out/rz_b1_gaps.err; head -14 gpurun_out/rz_b1_gaps.md | cut -c1-200; rm -rf gpurun_out/rz_b1; fi
echo "== GPU suite"
timeout 900 python -m pytest tests -m gpu -q --durations=5 -p no:cacheprovider < /dev/null > gpurun_out/rz_pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/rz_pytest_gpu.log; tail -9 gpurun_out/rz_pytest_gpu.log | cut -c1-200
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/rz_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/rz_smoke.log; tail -2 gpurun_out/rz_smoke.log | cut -c1-300
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 3 < /dev/null > gpurun_out/rz_bench.json 2> gpurun_out/rz_bench.err; echo "bench exit $?"
python tools/bench_brief.py gpurun_out/rz_bench.json bench | cut -c1-900
