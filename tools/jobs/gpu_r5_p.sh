#!/bin/bash
# round 5, call P: attention_h3 prologue with the image-scale scan under the first tile's DMA (library A/B: base = before, pro = after):
# kernel tests, one image per call (attention per launch, forward), then the headline step
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_long_sequences.py -m gpu -q -k attention_h3 < /dev/null 2>&1 | tail -1 | tee gpurun_out/r5p_attention_prologue.log
cp anyloc_amd/libanyloc_hip.so /tmp/lib_orig.so
for rep in 1 2; do for v in base pro; do
cp tools/ab_libs/lib_$v.so anyloc_amd/libanyloc_hip.so
python - $v <<'P' 2>&1 | grep -v "amdgpu.ids\|Seed set" | tee -a gpurun_out/r5p_attention_prologue.log
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from anyloc_amd import ops, synth, weights
import utilities
name = "dinov2_vitg14"
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device="cuda", depth=40))
ext = utilities.DinoV2ExtractFeatures(name, 31, "value", device="cuda")
for hw, B in (((322, 322), 1), ((322, 322), 2), ((476, 630), 1)):
    img = torch.randn(B, 3, *hw, device="cuda")
    for _ in range(5): tok = ext(img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): tok = ext(img)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 30
    ops.profile_enable(True); ops.profile_reset()
    for _ in range(10): ext(img)
    torch.cuda.synchronize(); ops.profile_enable(False)
    p = ops.profile_dump()["attention"]
    print(f"{sys.argv[1]}: B={B} {hw[0]}x{hw[1]}: attention {p['ms'] / p['calls'] * 1e3:.1f} us per launch, forward {wall * 1e3:.3f} ms", flush=True)
P
done; done
cp /tmp/lib_orig.so anyloc_amd/libanyloc_hip.so
REPS=2 STEPS=10 bash tools/gpu_ab_libs.sh base pro 2>&1 | cut -c1-330 | tee -a gpurun_out/r5p_attention_prologue.log
