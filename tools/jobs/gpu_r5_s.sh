#!/bin/bash
# round 5, call S: dispatch timestamps of the one-image forward of the final build (rocprofv3 --kernel-trace of tools/b1_trace_target.py,
# no HIP events) -> per-kernel time and the gaps between dependent launches (tools/b1_gaps.py), as round 4's r04_b1_kernel_trace_gaps.md
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r5s_b1 -o b1 -- python $R/tools/b1_trace_target.py 12 < /dev/null > $R/gpurun_out/r5s_b1_target.json 2> $R/gpurun_out/r5s_b1_target.err
cd $R
f=$(find gpurun_out/r5s_b1 -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python tools/b1_gaps.py "$f" 11 > gpurun_out/r5s_b1_gaps.md 2> gpurun_out/r5s_b1_gaps.err; head -30 gpurun_out/r5s_b1_gaps.md | cut -c1-260; rm -rf gpurun_out/r5s_b1; fi
cat gpurun_out/r5s_b1_target.json | tail -2 | cut -c1-300
