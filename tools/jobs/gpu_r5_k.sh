#!/bin/bash
# round 5, call K: the bench line of the final tree (VLAD stages on the pipeline's own tokens added) + the two-rank bench test
mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/r5k_bench.json 2> gpurun_out/r5k_bench.err; echo "bench exit $?"
tail -2 gpurun_out/r5k_bench.err | cut -c1-300
python - <<'P'
import json
d = json.loads(open("gpurun_out/r5k_bench.json").read().strip().splitlines()[-1])
st = d["stages"]
print("value", d["value"], "frac", d["roofline"]["frac"], "W", d["roofline"]["power"]["avg_w"], "sclk", d["roofline"]["power"]["sclk_mhz_avg"])
print({k: (st[k]["kernel_ms"], st[k]["frac"], st[k]["oracle_ok"]) for k in st if k.startswith("vlad")})
print("checks", d["roofline"]["checks"]["stages"])
P
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -q -k "bench" < /dev/null 2>&1 | tail -3
