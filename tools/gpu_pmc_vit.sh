#!/bin/bash
# rocprofv3 --pmc passes over the real ViT-g forward at the bench batch (tools/pmc_target_vit.py) -> per-kernel tables and
# profiles/pmc_traffic.json (the `roofline.traffic` source of bench.py).  Counters in their own runs, --kernel-trace only.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${ANYLOC_GEMM:-h3}
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/pmc_sq.log 2>&1
cd $R
for d in pmc_fetch pmc_write pmc_sq; do python tools/pmc_summarize.py gpurun_out/$d > gpurun_out/$d.md 2>&1; done
grep -E "attention|gemm_h3|gemm_x6|gemm_nt|layernorm" gpurun_out/pmc_sq.md | cut -c1-330
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write $MODE gpurun_out/pmc_traffic.json | grep -E "vit_|refetch|bytes_per" | cut -c1-120
