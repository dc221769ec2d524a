"""The fused VLAD launch with the shifted accumulation (option vlad_shift = 1) against the per-token centre gather (0),
interleaved, at 61 / 256 / 1024 images of 529 x 1536 tokens, K = 32: per-kernel HIP-event times (shorter of several
profiled calls) and wall time per call; random cluster membership and spatially coherent labels (runs of 8 equal labels).

    python tools/time_vlad_shift.py > gpurun_out/vlad_shift.log"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, synth  # noqa: E402

dev = "cuda"
c = 0.8 * synth.clustered_tokens(1, 32, 1536, n_modes=32, seed=3, device=dev)[0]


def kernel_ms(fn, reps=6):
    best = {}
    for _ in range(reps):
        ops.profile_enable(True)
        ops.profile_reset()
        fn()
        torch.cuda.synchronize()
        ops.profile_enable(False)
        for k, v in ops.profile_dump().items():
            best[k] = min(best.get(k, 1e9), v["ms"])
    return best


for n_img in (61, 256, 1024):
    toks = synth.clustered_tokens(n_img, 529, 1536, n_modes=32, seed=11, noise=0.6, device=dev)
    coherent = toks.reshape(n_img, 529, 1536)[:, torch.arange(529, device=dev) // 8 * 8]        # runs of 8 equal tokens -> equal labels
    for name, x in (("random", toks), ("runs of 8", coherent.contiguous())):
        byt = n_img * (529 * 1536 + 2 * 32 * 1536) * 4
        for rnd in range(2):
            for shift in (1, 0):
                with ops.options(vlad_shift=shift):
                    for _ in range(3):
                        ops.vlad(x, c)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        ops.vlad(x, c)
                    torch.cuda.synchronize()
                    wall = (time.perf_counter() - t0) / 20 * 1e3
                    k = kernel_ms(lambda: ops.vlad(x, c))
                f = k.get("vlad_fused", 0.0)
                print(f"{n_img:5d} images, {name:9s}, vlad_shift={shift}: vlad_fused {f * 1e3:7.1f} us = {byt / f / 1e9:6.2f} TB/s "
                      f"({byt / f / 1e9 / 8.0:.3f} of 8), all kernels of the call {sum(k.values()) * 1e3:7.1f} us, wall {wall * 1e3:7.1f} us  {k}",
                      flush=True)
    a = ops.vlad(toks, c)
    with ops.options(vlad_shift=0):
        b = ops.vlad(toks, c)
    print(f"      max rel L2 difference shift vs gather: {float(((a - b).norm(dim=1) / b.norm(dim=1)).max()):.2e}", flush=True)
