#!/bin/bash
# round 6, call D: FFN-bound telemetry with plain-store slots for few rows (the device-scope atomics of one image's 530 row words queue
# in one or two memory channels: vitg_b1 17.9 ms in call C) -- the telemetry tests, one image per call timed with the check on / off and
# with the atomics forced, then the bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_round6.py -x -q > gpurun_out/r6d_pytest.log 2>&1; tail -3 gpurun_out/r6d_pytest.log
python - <<'P' 2>&1 | grep -v "amdgpu.ids\|Seed set" | tee gpurun_out/r6d_b1_telemetry.log
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from anyloc_amd import ops, synth, weights
import utilities
name = "dinov2_vitg14"
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device="cuda", depth=40))
ext = utilities.DinoV2ExtractFeatures(name, 31, "value", device="cuda")
def t(img, n=40):
    for _ in range(5): ext(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ext(img)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for hw, B in (((322, 322), 1), ((476, 630), 1), ((322, 322), 61)):
    img = torch.randn(B, 3, *hw, device="cuda")
    res = {}
    for rep in range(2):
        for label, check, layout in (("check off", False, -1), ("slots", True, 0), ("atomics", True, 1), ("auto", True, -1)):
            ext.dino_model.ffn_check = check
            with ops.options(ffn_telem_atomic=layout):
                res.setdefault(label, []).append(round(t(img, 40 if B == 1 else 6), 3))
    print(f"B={B} {hw[0]}x{hw[1]} ms per call:", res, flush=True)
P
timeout 1200 python bench.py --steps 20 --warmup 5 < /dev/null > gpurun_out/r6d_bench.json 2> gpurun_out/r6d_bench.err; echo "bench exit $?"
python tools/bench_brief.py gpurun_out/r6d_bench.json bench | cut -c1-600 | head -3
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r6d_bench.json") if l.startswith('{"metric"')][0])
r = d["roofline"]
print({k: v for k, v in r.items() if not isinstance(v, (dict, list, str))})
P
