#!/bin/bash
# round 5, call B: the shifted-accumulation VLAD kernel -- its tests first, then the interleaved A/B timing, then the PMC
# FETCH_SIZE pass over the VLAD-mode launches; the long-sequence tests again (call A stopped at the first failure)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_vlad_cache.py tests/test_gpu_property.py -m gpu -q -x -s -k "vlad or VLAD" < /dev/null > gpurun_out/r5b_pytest_vlad.log 2>&1
echo "pytest(vlad) exit: $?" >> gpurun_out/r5b_pytest_vlad.log; grep -E "tight|passed|failed|rror|exit" gpurun_out/r5b_pytest_vlad.log | tail -30 | cut -c1-250
timeout 300 python tools/time_vlad_shift.py < /dev/null > gpurun_out/r5b_vlad_shift.log 2>&1; cut -c1-260 gpurun_out/r5b_vlad_shift.log | tail -30
timeout 1500 python -m pytest tests/test_gpu_long_sequences.py tests/test_gpu_vit.py tests/test_gpu_round4.py tests/test_abi.py tests/test_c_abi_host.py -m gpu -q --durations=8 -s < /dev/null > gpurun_out/r5b_pytest_new.log 2>&1
echo "pytest(new) exit: $?" >> gpurun_out/r5b_pytest_new.log; grep -E "token err|passed|failed|rror|exit" gpurun_out/r5b_pytest_new.log | tail -40 | cut -c1-220
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_vlad -o k -- python $R/tools/pmc_target_vlad.py < /dev/null > $R/gpurun_out/r5b_pmc_vlad.log 2>&1
cd $R
timeout 30 python tools/pmc_summarize.py gpurun_out/pmc_vlad --skip 0 < /dev/null > gpurun_out/r5b_pmc_vlad_fetch.md 2>&1
cat gpurun_out/r5b_pmc_vlad_fetch.md | cut -c1-200
# plan sweep of the block GEMMs at the scripts' default image shape (476 x 630: 1531 token rows at B = 1), 8-block model
timeout 600 python tools/sweep_b1.py 1 0,1,2,4,5,6,7 476x630 8 < /dev/null > gpurun_out/r5b_b1_480_plan_sweep.log 2>&1
grep -E "BEST|default plans|round-3" gpurun_out/r5b_b1_480_plan_sweep.log | cut -c1-400
