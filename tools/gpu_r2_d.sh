#!/bin/bash
# round 2: few-query retrieval bring-up
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_fullsize_properties.py -m gpu -q -x -k "topk or retrieval or recall" > gpurun_out/r2_topk.log 2>&1; tail -15 gpurun_out/r2_topk.log | cut -c1-250
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-modes > gpurun_out/r2_bench_topk.json 2> gpurun_out/r2_bench_topk.err
python tools/bench_brief.py gpurun_out/r2_bench_topk.json topk
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r2_bench_topk.json") if l.startswith("{")][-1])
print({k: v for k, v in d["roofline"]["kernels_ms_per_step"].items() if not k.startswith("vit_")})
PY
