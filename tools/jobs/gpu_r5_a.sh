#!/bin/bash
# round 5, call A: the new long-sequence parity tests + split-K stress first (fast feedback), hwmon probe for the power sampler,
# then the full bench line with the new stages (vitg_b1_480x640, generate_multi legs, config2_full_job, config3_whole_db, power)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
{ ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>&1; for h in /sys/class/drm/card*/device/hwmon/hwmon*; do ls $h; for f in power1_average power1_input power1_cap power1_cap_max freq1_input freq2_input; do [ -e $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"; done; done; rocm-smi --showpower --showmaxpower --json 2>&1 | head -5; } > gpurun_out/r5_hwmon_probe.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_long_sequences.py tests/test_gpu_vit.py -m gpu -q -x --durations=8 -s < /dev/null > gpurun_out/r5a_pytest_new.log 2>&1
echo "pytest(new) exit: $?" >> gpurun_out/r5a_pytest_new.log; grep -E "token err|passed|failed|error|exit" gpurun_out/r5a_pytest_new.log | tail -40 | cut -c1-220
timeout 900 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/r5a_bench.json 2> gpurun_out/r5a_bench.err; echo "bench exit $?"
tail -3 gpurun_out/r5a_bench.err | cut -c1-300
python tools/bench_brief.py gpurun_out/r5a_bench.json bench | cut -c1-1500
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r5a_bench.json").read().strip().splitlines()[-1])
    st = d.get("stages", {})
    print("power:", d["roofline"].get("power"))
    for k in ("vitg_b1", "vitg_b1_480x640"):
        v = st.get(k, {}); print(k, v.get("ms_per_image"), v.get("oracle_ok"), v.get("power"), list(v.get("kernels_ms", {}).items())[:8])
    sp = st.get("script_path_vitg", {}); print("script", sp.get("images_per_s"), sp.get("legs_ms"))
    print("config2_full_job", json.dumps(st.get("config2_full_job"))[:900])
    print("config3_whole_db", json.dumps(st.get("config3_whole_db"))[:600])
    print("vlad", {k: (st[k].get("kernel_ms"), st[k].get("frac")) for k in st if k.startswith("vlad")})
    print("rccl", d.get("rccl"))
except Exception as e:
    print("summary failed:", e)
P
