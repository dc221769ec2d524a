#!/bin/bash
# round 5, call D: where the fused VLAD launch spends its non-tile time (tokens per image varied); does polling the GPU's hwmon
# nodes cost the power-limited step anything (sampling period varied, over untimed steps); the 1100 < M <= 1700 plans; bench
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python tools/probe_vlad_fixed.py < /dev/null > gpurun_out/r5d_vlad_fixed_cost.log 2>&1; cut -c1-200 gpurun_out/r5d_vlad_fixed_cost.log | tail -22
for per in 0.02 0.25 1.0; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-modes --no-stages --no-cpu-baseline --power-steps 10 --power-period $per < /dev/null > gpurun_out/r5d_power_$per.json 2> gpurun_out/r5d_power_$per.err
  python - <<P
import json
d = json.loads(open("gpurun_out/r5d_power_$per.json").read().strip().splitlines()[-1])
p = d["roofline"]["power"]
print("period $per: value", d["value"], "ms/step", d["ms_per_step"], "while sampling", p.get("ms_per_step_while_sampling"), "W", p.get("avg_w"), "sclk", p.get("sclk_mhz_avg"), "samples", p.get("samples"))
P
done
timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_long_sequences.py -m gpu -q -x -k "small_m or split_k or vitg_extractor or other_facets" < /dev/null > gpurun_out/r5d_pytest_plans.log 2>&1
echo "pytest(plans) exit: $?" >> gpurun_out/r5d_pytest_plans.log; tail -4 gpurun_out/r5d_pytest_plans.log | cut -c1-200
timeout 900 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/r5d_bench.json 2> gpurun_out/r5d_bench.err; echo "bench exit $?"
tail -3 gpurun_out/r5d_bench.err | cut -c1-300
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r5d_bench.json").read().strip().splitlines()[-1])
    st = d.get("stages", {})
    print("value", d["value"], "frac", d["roofline"]["frac"], "power:", d["roofline"].get("power"))
    for k in ("vitg_b1", "vitg_b1_480x640"):
        v = st.get(k, {}); print(k, v.get("ms_per_image"), v.get("oracle_ok"), v.get("power"), list(v.get("kernels_ms", {}).items())[:7])
    sp = st.get("script_path_vitg", {}); print("script", sp.get("images_per_s"), sp.get("legs_ms"))
    print("config2_full_job", json.dumps(st.get("config2_full_job"))[:700])
    print("vlad", {k: (st[k].get("kernel_ms"), st[k].get("call_kernels_ms"), st[k].get("frac"), st[k].get("oracle_ok")) for k in st if k.startswith("vlad")})
    print("stages ok:", {k: v.get("oracle_ok") for k, v in st.items()})
except Exception as e:
    print("summary failed:", e)
P
