"""Small-batch forwards (ViT-g/14, 322x322, L31 value): sha256 of the tokens and ms per batch for B = 1, 2, 3 -- run once
with the defaults and once with ANYLOC_H3_DEEP_MAX=0 ANYLOC_LN_SMALL_ROWS=0 (the pre-change kernels): the hashes must agree
(the deep-stage GEMM and the one-row-per-wave LayerNorm change scheduling, not arithmetic)."""
import hashlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import synth, weights  # noqa: E402

import utilities  # noqa: E402

dev = "cuda"
name = sys.argv[1] if len(sys.argv) > 1 else "dinov2_vitg14"
layer = {"dinov2_vitg14": 31, "dinov2_vitl14": 23, "dinov2_vitb14": 11, "dinov2_vits14": 9}[name]
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device=dev, depth=layer + 1))
ext = utilities.DinoV2ExtractFeatures(name, layer, "value", device=dev)
g = torch.Generator(device=dev)
g.manual_seed(3)
for B in (1, 2, 3):
    img = torch.randn(B, 3, 322, 322, generator=g, device=dev)
    out = ext(img)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        ext(img)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{name} B={B} sha={h} finite={bool(torch.isfinite(out).all())} ms={ms:.2f}", flush=True)
