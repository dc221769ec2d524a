#!/bin/bash
# round 6, call E: where the per-call FFN check spends its host time at one image per call (cProfile of 60 calls)
mkdir -p gpurun_out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python - <<'P' 2>&1 | grep -v "amdgpu.ids\|Seed set" | tee gpurun_out/r6e_b1_check_profile.log
import cProfile, pstats, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from anyloc_amd import ops, synth, weights
import utilities
name = "dinov2_vitg14"
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device="cuda", depth=40))
ext = utilities.DinoV2ExtractFeatures(name, 31, "value", device="cuda")
img = torch.randn(1, 3, 322, 322, device="cuda")
for _ in range(5): ext(img)
torch.cuda.synchronize()
m = ext.dino_model
print("looseness of the last call:", m.ffn_looseness, "reruns", m.ffn_reruns, "exact", m.ffn_exact_blocks)
pr = cProfile.Profile()
pr.enable()
for _ in range(60): ext(img)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
P
