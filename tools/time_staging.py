"""Host <-> device copy rates from pageable memory at the sizes the reference scripts move: one image, one image's tokens,
256 images' tokens, the 10 000-row VLAD database (what anyloc_amd/ops.py (to_device / to_host) does: plain tensor.to() / .cpu(); the pinned-ring
variant of round 4 is in profiles/r04_staging.log).     python tools/time_staging.py > gpurun_out/staging.log"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops as staging  # noqa: E402  (the two copies live in ops.py since round 5)

dev = torch.device("cuda", 0)
for name, shape in (("image 3x322x322", (3, 322, 322)), ("tokens 529x1536", (529, 1536)), ("256 x tokens", (256, 529, 1536)),
                    ("database 10000x49152", (10000, 49152))):
    t = torch.randn(shape)
    nb = t.numel() * 4
    reps = 20 if nb < (64 << 20) else 3
    d = staging.to_device(t, dev); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        d = staging.to_device(t, dev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"H2D {name:22s} {dt*1e3:9.3f} ms  {nb/dt/1e9:6.2f} GB/s", flush=True)
    staging.to_host(d)
    t0 = time.perf_counter()
    for _ in range(reps):
        h = staging.to_host(d)
    dt = (time.perf_counter() - t0) / reps
    print(f"D2H {name:22s} {dt*1e3:9.3f} ms  {nb/dt/1e9:6.2f} GB/s", flush=True)
    del t, d, h
