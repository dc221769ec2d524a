#!/bin/bash
# round 6, call B: the gather hazard narrowed down (variants 9-11: accumulator through src1, one / two wait states; the shipped gather
# now carries `s_nop 3` behind every mode switch), the stand-alone probe tools/micro/gpr_idx_hazard.hip, the new tests (prepared flat
# index, overlapped sharded step, per-image FFN decision), then the whole GPU suite
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python tools/stress_vlad.py 20 --variants 0,1,2,9,10,11 > gpurun_out/r6b_vlad_variants.log 2>&1
tail -2 gpurun_out/r6b_vlad_variants.log
timeout 300 build/micro/gih 600 50 > gpurun_out/r6b_gpr_idx_probe.log 2>&1
cat gpurun_out/r6b_gpr_idx_probe.log
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_vit.py -x -q -s > gpurun_out/r6b_pytest_new.log 2>&1
tail -8 gpurun_out/r6b_pytest_new.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r6b_pytest.log 2>&1
tail -8 gpurun_out/r6b_pytest.log
