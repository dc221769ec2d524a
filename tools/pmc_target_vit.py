"""rocprofv3 --pmc target: the real ViT-g forward at the bench batch (B=61, 322x322) truncated to a few blocks,
so every block kernel (qkv / proj / w12 / fc2 GEMMs, attention, LayerNorm, quantisers) appears with its real
operands and neighbours.  usage: python tools/pmc_target_vit.py [depth=3] [reps=3] [batch=61]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import synth, weights  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 61
dev = "cuda"
sd = synth.synthetic_state_dict("dinov2_vitg14", 0, device=dev, depth=depth)
weights.register_state_dict("dinov2_vitg14", sd)
import utilities  # noqa: E402
ext = utilities.DinoV2ExtractFeatures("dinov2_vitg14", depth - 1, "token", device=dev)
img = torch.randn(B, 3, 322, 322, device=dev)
for _ in range(reps):
    out = ext(img)
torch.cuda.synchronize()
print("ok", tuple(out.shape), os.environ.get("ANYLOC_GEMM", "default"))
