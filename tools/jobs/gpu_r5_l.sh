#!/bin/bash
# round 5, call L: attention_h3 with two waves of 64 queries per workgroup (option attn_h3_qg = 2) against the default four waves of 32:
# the kernel test in both shapes, then the interleaved A/B on the headline workload and at one image per call
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "attention_h3" < /dev/null 2>&1 | tail -3
REPS=2 STEPS=10 bash tools/gpu_ab.sh "attn_h3_qg=1" "attn_h3_qg=2" 2>&1 | cut -c1-330 | tee gpurun_out/r5l_attention_qg2.log
python - <<'P' | tee -a gpurun_out/r5l_attention_qg2.log
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from anyloc_amd import ops, synth, weights
import utilities
name = "dinov2_vitg14"
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device="cuda", depth=8))
ext = utilities.DinoV2ExtractFeatures(name, 7, "value", device="cuda")
for hw in ((322, 322), (476, 630)):
    img = torch.randn(1, 3, *hw, device="cuda")
    for rep in range(2):
        for qg in (1, 2):
            with ops.options(attn_h3_qg=qg):
                for _ in range(3): ext(img)
                ops.profile_enable(True); ops.profile_reset()
                for _ in range(10): ext(img)
                torch.cuda.synchronize(); ops.profile_enable(False)
                p = ops.profile_dump()["attention"]
                print(f"B=1 {hw[0]}x{hw[1]} attn_h3_qg={qg}: attention {p['ms'] / p['calls'] * 1e3:.1f} us per launch")
P
