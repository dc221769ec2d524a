#!/bin/bash
# rocprofv3 --pmc passes over the screened retrieval (tools/pmc_target_screen.py) -> per-kernel tables.  Counters in their own
# runs, --kernel-trace only.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_fetch -o k -- python $R/tools/pmc_target_screen.py > $R/gpurun_out/pmcs_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_write -o k -- python $R/tools/pmc_target_screen.py > $R/gpurun_out/pmcs_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_sq -o k -- python $R/tools/pmc_target_screen.py > $R/gpurun_out/pmcs_sq.log 2>&1
cd $R
for d in pmcs_fetch pmcs_write pmcs_sq; do python tools/pmc_summarize.py gpurun_out/$d > gpurun_out/$d.md 2>&1; rm -rf gpurun_out/$d; done
grep -E "screen|merge|split" gpurun_out/pmcs_fetch.md gpurun_out/pmcs_write.md gpurun_out/pmcs_sq.md | cut -c1-400
