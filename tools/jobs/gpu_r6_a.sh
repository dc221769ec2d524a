#!/bin/bash
# round 6, call A: the GPU suite on the ABI-8 build (per-image FFN-bound decision, per-device LDS attribute, workspace for a caller's parts
# count) + the gather-variant stress of the one-pass VLAD kernel (hazard study: variants compiled into one library, option vlad_gather_v)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/stress_vlad.py 20 --variants 0,1,2,3,4,5,7,8 > gpurun_out/r6a_vlad_variants.log 2>&1
tail -3 gpurun_out/r6a_vlad_variants.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6a_pytest.log 2>&1
tail -5 gpurun_out/r6a_pytest.log
