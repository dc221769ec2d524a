#!/bin/bash
# gpurun: GEMM tile-configuration sweep, rocprofv3 kernel trace of the bench, PMC counters of the GEMMs.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in 0 1 2 3 4; do
  ANYLOC_GEMM_CFG=$c timeout 300 python tools/microbench_gemm.py 32 2>&1 | tail -1
done | tee gpurun_out/gemm_sweep.log
timeout 600 python -m pytest tests -m gpu -q -x -k "vlad_hard_vs_oracle or gemm" > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -o gemm -- python $R/tools/microbench_gemm.py 32 > $R/gpurun_out/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o gemm -- python $R/tools/microbench_gemm.py 32 > $R/gpurun_out/pmc_fetch.log 2>&1
cd $R
find gpurun_out -name "*.csv" | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); echo "== $f"; head -30 "$f" | cut -c1-250
