"""Round 6 probe: what the FIRST ``VLAD.generate_multi`` on a CPU tensor of patch descriptors costs (the reference script's
call, ``scripts/dino_v2_vlad.py:236-260``: 256 images x 529 x 1536 fp32 = 832 MB handed over once) as a function of the
piece size the host path streams it in, cold (allocator caches and library workspaces released first) and warm; and what the
result's way back costs as ``.cpu()`` against a copy into pinned memory.

    python tools/probe_host_staging.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import utilities  # noqa: E402
from anyloc_amd import _lib, synth  # noqa: E402

dev = "cuda"
n_img, N, D, K = 256, 529, 1536, 32
toks = synth.clustered_tokens(n_img, N, D, n_modes=K, seed=11, noise=0.6, device=dev).cpu()
c = 0.8 * synth.clustered_tokens(1, K, D, n_modes=K, seed=3, device=dev)[0]
v = utilities.VLAD(K, D, cache_dir=None)
v.c_centers = c.cpu()                     # (a fitted object: only the centres matter to generate_multi)
v.kmeans = object()


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, r


ref = None
for mb in (256, 128, 64, 32, 16):
    v.HOST_CHUNK_BYTES = mb << 20
    cold = []
    for rep in range(3):
        _lib.release_workspaces()
        torch.cuda.empty_cache()
        ms, out = timed(lambda: v.generate_multi(toks))
        cold.append(ms)
        if ref is None:
            ref = out
        assert torch.equal(out, ref), "piece size changed the bits"
    warm = [timed(lambda: v.generate_multi(toks))[0] for _ in range(3)]
    print(f"pieces of {mb:4d} MB: cold {min(cold):7.2f} .. {max(cold):7.2f} ms   warm {min(warm):6.2f} .. {max(warm):6.2f} ms", flush=True)

# the result's way back (50 MB) and a one-image token tensor's (3.25 MB)
for shape in ((256, K * D), (1, N, D)):
    x = torch.randn(*shape, device=dev)
    a = [timed(lambda: x.cpu())[0] for _ in range(4)]

    def pinned():
        h = torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
        h.copy_(x)
        return h
    b = [timed(pinned)[0] for _ in range(4)]
    keep = [pinned() for _ in range(2)]                # (blocks held: the caching host allocator cannot hand them out again)
    b2 = [timed(pinned)[0] for _ in range(2)]
    print(f"{x.numel() * 4 / 1e6:6.2f} MB device -> host: .cpu() {['%.2f' % t for t in a]} ms; into fresh pinned memory {['%.2f' % t for t in b]} "
          f"(two more while earlier results are alive: {['%.2f' % t for t in b2]}) ms", flush=True)
    del keep
