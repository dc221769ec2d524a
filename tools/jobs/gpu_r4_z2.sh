#!/bin/bash
# round 4, follow-up of gpu_r4_z.sh: the one-image forward's dispatch-timestamp trace again (the first analysis cut the
# periods wrongly and the trace had been deleted), the same loop with the kernel arguments in device memory
# (HIP_FORCE_DEV_KERNARG=1), and the clocks / power the chip reports while it runs one-image forwards.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
echo "== plain"; timeout 200 python tools/b1_trace_target.py 100 2> gpurun_out/rz2_plain.err | tee gpurun_out/rz2_b1_plain.json
echo "== HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 timeout 200 python tools/b1_trace_target.py 100 2> gpurun_out/rz2_kernarg.err | tee gpurun_out/rz2_b1_dev_kernarg.json
echo "== HIP_FORCE_DEV_KERNARG=0"; HIP_FORCE_DEV_KERNARG=0 timeout 200 python tools/b1_trace_target.py 100 2> /dev/null | tee gpurun_out/rz2_b1_host_kernarg.json
echo "== rocprofv3 --kernel-trace"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rz2_b1 -o b1 -- python $R/tools/b1_trace_target.py 12 < /dev/null > $R/gpurun_out/rz2_b1_target.json 2> $R/gpurun_out/rz2_b1_target.err
cd $R; cat gpurun_out/rz2_b1_target.json
f=$(find gpurun_out/rz2_b1 -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then
  python tools/b1_gaps.py "$f" 11 > gpurun_out/rz2_b1_gaps.md 2> gpurun_out/rz2_b1_gaps.err; head -16 gpurun_out/rz2_b1_gaps.md | cut -c1-260
  python - "$f" <<'PY'
import csv, gzip, sys
rows = list(csv.DictReader(open(sys.argv[1], newline="")))
with gzip.open("gpurun_out/rz2_b1_kernel_trace_min.csv.gz", "wt", newline="") as fh:
    w = csv.writer(fh); w.writerow(["Start_Timestamp", "End_Timestamp", "Kernel_Name"])
    for r in rows: w.writerow([r["Start_Timestamp"], r["End_Timestamp"], r["Kernel_Name"]])
PY
  rm -rf gpurun_out/rz2_b1
fi
echo "== clocks while one-image forwards run"
timeout 120 python tools/b1_trace_target.py 1500 > gpurun_out/rz2_b1_long.json 2> /dev/null &
sleep 22
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" | head -8; echo --; sleep 1.5; done | tee gpurun_out/rz2_clocks_b1.log
wait
cat gpurun_out/rz2_b1_long.json
