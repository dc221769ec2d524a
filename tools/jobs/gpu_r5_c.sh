#!/bin/bash
# round 5, call C: the compact (runtime-loop) epilogue of the fused VLAD kernel: VLAD tests, the shift / gather A/B again, the whole GPU
# suite, then the bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_vlad_cache.py tests/test_gpu_property.py tests/test_gpu_round4.py -m gpu -q -x -s -k "vlad or VLAD or cpu_tensor" < /dev/null > gpurun_out/r5c_pytest_vlad.log 2>&1
echo "pytest(vlad) exit: $?" >> gpurun_out/r5c_pytest_vlad.log; grep -E "tight|passed|failed|rror|exit" gpurun_out/r5c_pytest_vlad.log | tail -30 | cut -c1-250
timeout 300 python tools/time_vlad_shift.py < /dev/null > gpurun_out/r5c_vlad_shift.log 2>&1; cut -c1-200 gpurun_out/r5c_vlad_shift.log | tail -28
timeout 1800 python -m pytest tests -m gpu -q --durations=6 < /dev/null > gpurun_out/r5c_pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r5c_pytest_gpu.log; tail -14 gpurun_out/r5c_pytest_gpu.log | cut -c1-220
timeout 900 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/r5c_bench.json 2> gpurun_out/r5c_bench.err; echo "bench exit $?"
tail -3 gpurun_out/r5c_bench.err | cut -c1-300
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r5c_bench.json").read().strip().splitlines()[-1])
    st = d.get("stages", {})
    print("value", d["value"], "frac", d["roofline"]["frac"], "power:", d["roofline"].get("power"))
    for k in ("vitg_b1", "vitg_b1_480x640"):
        v = st.get(k, {}); print(k, v.get("ms_per_image"), v.get("oracle_ok"), v.get("power"))
    sp = st.get("script_path_vitg", {}); print("script", sp.get("images_per_s"), sp.get("legs_ms"))
    print("config2_full_job", json.dumps(st.get("config2_full_job"))[:1100])
    print("config3_whole_db", json.dumps(st.get("config3_whole_db"))[:400])
    print("vlad", {k: (st[k].get("kernel_ms"), st[k].get("call_kernels_ms"), st[k].get("frac"), st[k].get("oracle_ok")) for k in st if k.startswith("vlad")})
    print("kmeans", st["kmeans_5Mx1536"]["frac"], "stages ok:", {k: v.get("oracle_ok") for k, v in st.items()})
except Exception as e:
    print("summary failed:", e)
P
