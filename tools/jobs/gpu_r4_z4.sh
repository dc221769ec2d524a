#!/bin/bash
# round 4, final validation of the build with the 192 x 128 / 6-deep one-image w12 plan as default: GPU suite + one-image wall time
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 230 python -m pytest tests -m gpu -q --durations=3 -p no:cacheprovider < /dev/null > gpurun_out/rz4_pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/rz4_pytest_gpu.log; tail -7 gpurun_out/rz4_pytest_gpu.log | cut -c1-200
timeout 60 python tools/b1_trace_target.py 100 2> /dev/null | tee gpurun_out/rz4_b1.json
