#!/bin/bash
# round 5, call H: the fused VLAD kernel trimmed for vector-instruction issue (shift-address fetch, division-free stash rows, fma
# residual, pipelined LDS reads; the shifted-accumulation variants are gone): VLAD tests, tokens-per-image sweep, VLAD stage,
# PMC FETCH_SIZE of the VLAD-mode launch, full suite, bench
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vlad_topk.py tests/test_gpu_vlad_cache.py tests/test_gpu_property.py tests/test_gpu_round4.py tests/test_c_abi_host.py -m gpu -q -x -k "vlad or VLAD or cpu_tensor or c_host" < /dev/null > gpurun_out/r5h_pytest_vlad.log 2>&1
echo "pytest(vlad) exit: $?" >> gpurun_out/r5h_pytest_vlad.log; tail -4 gpurun_out/r5h_pytest_vlad.log | cut -c1-250
timeout 300 python tools/probe_vlad_fixed.py < /dev/null > gpurun_out/r5h_vlad_fixed_cost.log 2>&1; cut -c1-200 gpurun_out/r5h_vlad_fixed_cost.log | tail -20
timeout 300 python tools/run_stage.py vlad_61img vlad_256img --check < /dev/null > gpurun_out/r5h_stage_vlad.json 2>&1; cut -c1-420 gpurun_out/r5h_stage_vlad.json | tail -3
cd /tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_vlad_h -o k -- python $R/tools/pmc_target_vlad.py < /dev/null > $R/gpurun_out/r5h_pmc_vlad.log 2>&1
cd $R
timeout 30 python tools/pmc_summarize.py gpurun_out/pmc_vlad_h --skip 1 < /dev/null > gpurun_out/r5h_pmc_vlad_fetch.md 2>&1
cat gpurun_out/r5h_pmc_vlad_fetch.md | cut -c1-200
timeout 1800 python -m pytest tests -m gpu -q --durations=6 < /dev/null > gpurun_out/r5h_pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/r5h_pytest_gpu.log; tail -12 gpurun_out/r5h_pytest_gpu.log | cut -c1-220
timeout 900 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/r5h_bench.json 2> gpurun_out/r5h_bench.err; echo "bench exit $?"
tail -3 gpurun_out/r5h_bench.err | cut -c1-300
python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r5h_bench.json").read().strip().splitlines()[-1])
    st = d.get("stages", {})
    print("value", d["value"], "frac", d["roofline"]["frac"], "power:", d["roofline"].get("power"))
    for k in ("vitg_b1", "vitg_b1_480x640"):
        v = st.get(k, {}); print(k, v.get("ms_per_image"), v.get("oracle_ok"), (v.get("power") or {}).get("avg_w"), list(v.get("kernels_ms", {}).items())[:7])
    sp = st.get("script_path_vitg", {}); print("script", sp.get("images_per_s"), sp.get("legs_ms"))
    print("config2_full_job", json.dumps(st.get("config2_full_job"))[:700])
    print("vlad", {k: (st[k].get("kernel_ms"), st[k].get("call_kernels_ms"), st[k].get("frac"), st[k].get("oracle_ok")) for k in st if k.startswith("vlad")})
    print("stages ok:", {k: v.get("oracle_ok") for k, v in st.items()})
except Exception as e:
    print("summary failed:", e)
P
