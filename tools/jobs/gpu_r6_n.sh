#!/bin/bash
# round 6, call N: one-shot screened search with the leading-plane-only quantiser; C host (screened pass); stages again
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_screen.py tests/test_c_abi_host.py tests/test_gpu_round6.py tests/test_gpu_vlad_topk.py -x -q -s > gpurun_out/r6n_pytest.log 2>&1
grep -i "topk\|passed\|failed" gpurun_out/r6n_pytest.log | tail -14
timeout 1200 python tools/time_screen.py > gpurun_out/r6n_time_screen.log 2>&1
grep -v amdgpu gpurun_out/r6n_time_screen.log | tail -14
timeout 1500 python tools/run_stage.py config3_whole_db > gpurun_out/r6n_stage_config3_whole_db.json 2> gpurun_out/r6n_stage_config3_whole_db.err
tail -c 900 gpurun_out/r6n_stage_config3_whole_db.json
