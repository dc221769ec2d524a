#!/bin/bash
# round 2: attention tuning check + PMC passes of the fused forward (traffic json) + PCA retest
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pca.py tests/test_gpu_kernels.py -m gpu -q -k "pca or attention" > gpurun_out/r2_newtests.log 2>&1; tail -4 gpurun_out/r2_newtests.log | cut -c1-200
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes > gpurun_out/r2_bench_attn.json 2> gpurun_out/r2_bench_attn.err
python tools/bench_brief.py gpurun_out/r2_bench_attn.json attn
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2f_pmc_fetch -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/r2f_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r2f_pmc_write -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/r2f_pmc_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/r2f_pmc_sq -o k -- python $R/tools/pmc_target_vit.py > $R/gpurun_out/r2f_pmc_sq.log 2>&1
cd $R
for d in r2f_pmc_fetch r2f_pmc_write r2f_pmc_sq; do python tools/pmc_summarize.py gpurun_out/$d > gpurun_out/$d.md 2>&1; done
grep -E "attention|gemm_h3|layernorm" gpurun_out/r2f_pmc_sq.md | cut -c1-330
grep -E "attention|gemm_h3|layernorm" gpurun_out/r2f_pmc_write.md | cut -c1-200
python tools/pmc_traffic.py gpurun_out/r2f_pmc_fetch gpurun_out/r2f_pmc_write h3 gpurun_out/pmc_traffic.json | grep -E "vit_|refetch|bytes_per" | cut -c1-120
