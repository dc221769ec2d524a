#!/bin/bash
# round 6, call C: the stand-alone hazard probe with idle issue slots in front of the mode switch (variants D / E), the round-6 tests
# again (FlatIndex follows the library's path rule), the host-staging probe (piece size of VLAD.generate_multi on a CPU tensor, pinned
# results), then the full bench line of the ABI-8 build
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 build/micro/gih 600 50 > gpurun_out/r6c_gpr_idx_probe.log 2>&1; cat gpurun_out/r6c_gpr_idx_probe.log
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q > gpurun_out/r6c_pytest_round6.log 2>&1; tail -3 gpurun_out/r6c_pytest_round6.log
timeout 600 python tools/probe_host_staging.py > gpurun_out/r6c_host_staging.log 2>&1; grep -v amdgpu.ids gpurun_out/r6c_host_staging.log | tail -12
timeout 1200 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/r6c_bench.json 2> gpurun_out/r6c_bench.err; echo "bench exit $?"
python tools/bench_brief.py gpurun_out/r6c_bench.json bench | cut -c1-900
