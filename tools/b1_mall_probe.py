"""Are the one-image block GEMMs waiting for their WEIGHTS?  ViT-G/14 at B = 1 streams 113 MB of weight images per block from
HBM (3.5 GB per forward: no reuse inside the 256 MB Infinity Cache).  A 2-block model (226 MB) keeps its weights on-die from one
forward to the next: the per-launch times of the same kernels with and without that residency say how much of a launch is
HBM latency.     python tools/b1_mall_probe.py > gpurun_out/b1_mall_probe.log"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, synth, weights  # noqa: E402

import utilities  # noqa: E402

dev = "cuda"
name = "dinov2_vitg14"
img = torch.randn(1, 3, 322, 322, device=dev)
for depth in (2, 32):
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device=dev, depth=depth))
    ext = utilities.DinoV2ExtractFeatures(name, depth - 1, "token", device=dev)
    ext.dino_model.ffn_check = False
    for _ in range(5):
        ext(img)
    torch.cuda.synchronize()
    n = 40 if depth == 2 else 6
    t0 = time.perf_counter()
    for _ in range(n):
        ext(img)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    ops.profile_enable(True)
    ops.profile_reset()
    for _ in range(n):
        ext(img)
    torch.cuda.synchronize()
    prof = ops.profile_dump()
    ops.profile_enable(False)
    print(f"depth {depth}: {wall * 1e3:.3f} ms per forward = {wall * 1e6 / depth:.1f} us per block;  per launch (us): " +
          "  ".join(f"{k}={v['ms'] / v['calls'] * 1e3:.1f}" for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:8]), flush=True)
    weights.unregister_state_dict(name)
    del ext
