#!/bin/bash
# round 6, call F: one image per call with the per-call FFN check read back through pinned memory + NumPy (call E: `.cpu()` + torch CPU
# reductions of the hundred bytes cost 15 + 9 ms per call on the 256-thread host)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python - <<'P' 2>&1 | grep -v "amdgpu.ids\|Seed set" | tee gpurun_out/r6f_b1_telemetry.log
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
def t_cpu(img, n=40):                       # the scripts' pattern: a result on the host per call
    for _ in range(5): ext(img).cpu()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ext(img).cpu()
    return (time.perf_counter() - t0) / n * 1e3
for hw, B in (((322, 322), 1), ((476, 630), 1), ((322, 322), 61)):
    img = torch.randn(B, 3, *hw, device="cuda")
    res = {}
    for rep in range(2):
        for label, check, layout in (("check off", False, -1), ("slots", True, 0), ("atomics", True, 1), ("auto", True, -1)):
            ext.dino_model.ffn_check = check
            with ops.options(ffn_telem_atomic=layout):
                res.setdefault(label, []).append(round(t(img, 40 if B == 1 else 6), 3))
                if B == 1: res.setdefault(label + " +.cpu()", []).append(round(t_cpu(img), 3))
    print(f"B={B} {hw[0]}x{hw[1]} ms per call:", res, flush=True)
P
timeout 600 python -m pytest tests/test_gpu_vit.py tests/test_gpu_round6.py -x -q 2>&1 | tail -2
