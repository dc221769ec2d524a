#!/bin/bash
# round 2, first GPU call: new full-size parity tests + PMC traffic of the h3 forward
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fullsize_parity.py tests/test_gpu_kernels.py tests/test_gpu_vlad_topk.py -m gpu -q -x -s --durations=8 > gpurun_out/r2_parity.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r2_parity.log; grep -E "token err|passed|failed|Error|assert" gpurun_out/r2_parity.log | cut -c1-220 | tail -40
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_fetch -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/r2_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_write -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/r2_pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/r2_pmc_sq -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/r2_pmc_sq.log 2>&1
cd $R
for d in r2_pmc_fetch r2_pmc_write r2_pmc_sq; do tail -1 gpurun_out/$d.log | cut -c1-200; python tools/pmc_summarize.py gpurun_out/$d > gpurun_out/$d.md 2>&1; cat gpurun_out/$d.md | cut -c1-400; done
