#!/bin/bash
# round 6, call K: the screened retrieval: tests, then the timing at configs[2]'s shard shape
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_screen.py -x -q -s > gpurun_out/r6k_pytest_screen.log 2>&1
tail -25 gpurun_out/r6k_pytest_screen.log
timeout 1200 python tools/time_screen.py > gpurun_out/r6k_time_screen.log 2>&1
tail -20 gpurun_out/r6k_time_screen.log
