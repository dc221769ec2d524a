#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention_h3" 2>&1 | tail -1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r2h_pmc_a -o k -- python $R/tools/pmc_target_attn.py > $R/gpurun_out/r2h_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/r2h_pmc_b -o k -- python $R/tools/pmc_target_attn.py > $R/gpurun_out/r2h_b.log 2>&1
cd $R
for d in r2h_pmc_a r2h_pmc_b; do python tools/pmc_summarize.py gpurun_out/$d --match attention_h3 | cut -c1-400; done
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes > gpurun_out/r2_bench_h.json 2> gpurun_out/r2_bench_h.err
python tools/bench_brief.py gpurun_out/r2_bench_h.json readlane
