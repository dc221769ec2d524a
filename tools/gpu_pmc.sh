#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o gemm -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq2 -o gemm -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_sq2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o gemm -- python $R/tools/pmc_target.py > $R/gpurun_out/pmc_fetch.log 2>&1
cd $R
for d in pmc_sq pmc_sq2 pmc_fetch; do tail -2 gpurun_out/$d.log | cut -c1-200; find gpurun_out/$d -name "*.csv" | head; done
