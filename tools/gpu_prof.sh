#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command (the summary that must agree with the bench's own HIP-event timing)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes > $R/gpurun_out/r2_prof_bench.json 2> $R/gpurun_out/r2_prof_bench.err
cd $R; f=$(find gpurun_out/r2_prof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r2_kernel_stats.csv; head -9 gpurun_out/r2_kernel_stats.csv | cut -c1-200
python tools/bench_brief.py gpurun_out/r2_prof_bench.json profiled
