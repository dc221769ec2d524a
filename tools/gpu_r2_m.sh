#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in 2 1; do
ANYLOC_KMEANS_FUSED_V=$v timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r2m_a$v -o k -- python $R/tools/pmc_target_kmeans.py > /dev/null 2>&1
ANYLOC_KMEANS_FUSED_V=$v timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2m_b$v -o k -- python $R/tools/pmc_target_kmeans.py > /dev/null 2>&1
python $R/tools/pmc_summarize.py $R/gpurun_out/r2m_a$v --match fused | cut -c1-330
python $R/tools/pmc_summarize.py $R/gpurun_out/r2m_b$v --match fused | cut -c1-330
done
