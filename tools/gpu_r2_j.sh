#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for gm in 8 4 16 32 2; do ANYLOC_H3_GM=$gm timeout 200 python tools/sweep_h3.py 2>&1 | grep "^{" | sed "s/^{/{\"gm\": $gm, /"; done | tee gpurun_out/r2_h3_gm_sweep.log
cd /tmp
for gm in 8 16; do
  ANYLOC_H3_GM=$gm timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r2j_fetch_gm$gm -o k -- python $R/tools/sweep_h3.py > /dev/null 2>&1
  python $R/tools/pmc_summarize.py $R/gpurun_out/r2j_fetch_gm$gm --match gemm_h3 | cut -c1-200
done
