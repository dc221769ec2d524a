#!/bin/bash
# round 6, call L: the screened retrieval as the default (option topk_screen = -1): its tests, then the whole GPU suite
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_screen.py -x -q -s > gpurun_out/r6l_pytest_screen.log 2>&1
tail -12 gpurun_out/r6l_pytest_screen.log
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6l_pytest_gpu.log 2>&1
tail -12 gpurun_out/r6l_pytest_gpu.log
