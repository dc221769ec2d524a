"""Extractor throughput vs batch size (ViT-G/14, 322x322, layer 31 'value'): the reference's scripts
call the extractor with B=1; bench.py uses B=61."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, synth, weights  # noqa: E402

import utilities  # noqa: E402

dev = "cuda"
name = "dinov2_vitg14"
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device=dev, depth=32))
ext = utilities.DinoV2ExtractFeatures(name, 31, "value", device=dev)
BATCHES = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 2, 4, 8, 16, 30, 61)
PROFILE = os.environ.get("ANYLOC_BATCH_PROFILE") == "1"          # per-kernel ms per forward (the library's HIP-event scopes)
for B in BATCHES:
    img = torch.randn(B, 3, 322, 322, device=dev)
    for _ in range(2):
        ext(img)
    torch.cuda.synchronize()
    n = max(2, 64 // B)
    t0 = time.perf_counter()
    for _ in range(n):
        ext(img)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:3d}: {dt*1e3:8.2f} ms/batch  {B/dt:7.1f} img/s  {B*0.9873/dt:6.1f} TFLOP/s", flush=True)
    if PROFILE:
        ops.profile_enable(True)
        ops.profile_reset()
        for _ in range(n):
            ext(img)
        torch.cuda.synchronize()
        prof = ops.profile_dump()
        ops.profile_enable(False)
        top = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:9]
        print("      " + "  ".join(f"{k}={v['ms'] / n:.3f}" for k, v in top), flush=True)
