#!/bin/bash
# rocprofv3 --pmc FETCH_SIZE over one k-means step at 2 M x 1536 (tools/pmc_target_kmeans.py): HBM-side read bytes of the
# fused kernel against the algorithmic 12.29 GB (tokens read once)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 110 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_km -o k -- python $R/tools/pmc_target_kmeans.py < /dev/null > $R/gpurun_out/pmc_km.log 2>&1
cd $R
timeout 20 python tools/pmc_summarize.py gpurun_out/pmc_km < /dev/null > gpurun_out/pmc_km.md 2>&1
grep -E "fused|reduce|kernel \|" gpurun_out/pmc_km.md | cut -c1-250
