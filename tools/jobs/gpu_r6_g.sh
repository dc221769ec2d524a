#!/bin/bash
# round 6, call G: validation of the build -- the GPU suite, smoke(), the full bench line, the rocprofv3 kernel stats of the bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 2400 python -m pytest tests -m gpu -q --durations=6 < /dev/null > gpurun_out/r6g_pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r6g_pytest_gpu.log; tail -12 gpurun_out/r6g_pytest_gpu.log | cut -c1-220
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/r6g_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/r6g_smoke.log; tail -2 gpurun_out/r6g_smoke.log
timeout 1200 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/r6g_bench.json 2> gpurun_out/r6g_bench.err; echo "bench exit $?"
python tools/bench_brief.py gpurun_out/r6g_bench.json bench | cut -c1-500 | head -2
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r6g_bench.json") if l.startswith('{"metric"')][0])
print({k: v for k, v in d["roofline"].items() if not isinstance(v, (dict, list, str))})
print(d["stages"]["script_path_vitg"]["legs_ms"])
P
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r6g_prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes --no-stages < /dev/null > $R/gpurun_out/r6g_prof_bench.json 2> $R/gpurun_out/r6g_prof_bench.err
cd $R; f=$(find gpurun_out/r6g_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r6g_kernel_stats.csv && head -8 gpurun_out/r6g_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/r6g_prof
python tools/bench_brief.py gpurun_out/r6g_prof_bench.json profiled | head -1 | cut -c1-400
