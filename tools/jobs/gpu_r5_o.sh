#!/bin/bash
# round 5, call O: attention_h3 with two query waves x two key waves per workgroup (attn_h3_ks): kernel tests in all shapes, then one image
# per call at 322 x 322 and 476 x 630 (and 2, 3 images) with the split forced off / on, attention per launch and the whole forward
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_long_sequences.py -m gpu -q -k "attention_h3" < /dev/null 2>&1 | grep -E "^E  |passed|failed" | head -20 | tee gpurun_out/r5o_attention_ks.log
python - <<'P' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5o_attention_ks.log
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from anyloc_amd import ops, synth, weights
import utilities
name = "dinov2_vitg14"
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device="cuda", depth=40))
ext = utilities.DinoV2ExtractFeatures(name, 31, "value", device="cuda")
for hw, batches in (((322, 322), (1, 2, 3, 4, 6)), ((476, 630), (1, 2))):
    for B in batches:
        img = torch.randn(B, 3, *hw, device="cuda")
        ref = None
        for rep in range(1):
            for ks in (1, 2):        # (3 = two query waves of 64 x two key waves existed for this call only: measured, removed)
                with ops.options(attn_h3_ks=ks):
                    for _ in range(3): tok = ext(img)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for _ in range(20): tok = ext(img)
                    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20
                    ops.profile_enable(True); ops.profile_reset()
                    for _ in range(10): ext(img)
                    torch.cuda.synchronize(); ops.profile_enable(False)
                    p = ops.profile_dump()["attention"]
                if ref is None: ref = tok
                print(f"B={B} {hw[0]}x{hw[1]} attn_h3_ks={ks}: attention {p['ms'] / p['calls'] * 1e3:.1f} us per launch, forward {wall * 1e3:.3f} ms, "
                      f"max |token difference to ks=1| {float((tok - ref).abs().max()):.2e}", flush=True)
P
