"""Host <-> device copy rates of the pinned staging path (anyloc_amd/staging.py) against plain tensor.to() / .cpu() from
pageable memory, at the sizes the reference scripts move: one image, one image's tokens, 256 images' tokens, the 10 000-row
VLAD database.     python tools/time_staging.py > gpurun_out/staging.log"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import staging  # noqa: E402

dev = torch.device("cuda", 0)
for name, shape in (("image 3x322x322", (3, 322, 322)), ("tokens 529x1536", (529, 1536)), ("256 x tokens", (256, 529, 1536)),
                    ("database 10000x49152", (10000, 49152))):
    t = torch.randn(shape)
    nb = t.numel() * 4
    reps = 20 if nb < (64 << 20) else 3
    for label, fn in (("tensor.to(device)", lambda: t.to(dev)), ("staging.to_device", lambda: staging.to_device(t, dev))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            d = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"H2D {name:22s} {label:18s} {dt*1e3:9.3f} ms  {nb/dt/1e9:6.2f} GB/s", flush=True)
    for label, fn in (("tensor.cpu()", lambda: d.cpu()), ("staging.to_host", lambda: staging.to_host(d))):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            h = fn()
        dt = (time.perf_counter() - t0) / reps
        print(f"D2H {name:22s} {label:18s} {dt*1e3:9.3f} ms  {nb/dt/1e9:6.2f} GB/s", flush=True)
    for th in (1, 4, 16):
        staging.COPY_THREADS = th
        staging._pool = None
        if nb >= (64 << 20):
            staging.to_device(t, dev); torch.cuda.synchronize()
            t0 = time.perf_counter(); staging.to_device(t, dev); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"H2D {name:22s} staging, {th:2d} threads {dt*1e3:9.3f} ms  {nb/dt/1e9:6.2f} GB/s", flush=True)
    staging.COPY_THREADS = 8
    staging._pool = None
    del t, d, h
