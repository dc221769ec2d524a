"""Where does the fused VLAD launch spend the time that is not tiles?  256 images (one workgroup each, parts forced to 1) of
N tokens x 1536, N = 16 ... 4232: kernel time against the number of 16-token tiles per workgroup -> slope (us per tile)
and intercept (us per workgroup); the same rows through the k-means step (the same kernel in k-means mode, no per-image
epilogue) for comparison -- a k-means chunk has at least 1 024 rows, so only N >= 1 058 gives it the same tiles per workgroup
and its slope is taken from those points.

    python tools/probe_vlad_fixed.py > gpurun_out/vlad_fixed_cost.log"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, synth  # noqa: E402

dev = "cuda"
c = 0.8 * synth.clustered_tokens(1, 32, 1536, n_modes=32, seed=3, device=dev)[0]


def kernel_us(fn, tag, reps=6):
    best = 1e9
    for _ in range(reps):
        ops.profile_enable(True)
        ops.profile_reset()
        fn()
        torch.cuda.synchronize()
        ops.profile_enable(False)
        best = min(best, ops.profile_dump()[tag]["ms"] * 1e3)
    return best


rows = []
for n_img in (256, 1024):
    for N in (16, 144, 272, 529, 1058, 2116, 4232):
        if n_img * N * 1536 * 4 > 30e9:
            continue
        toks = synth.clustered_tokens(n_img, N, 1536, n_modes=32, seed=11, noise=0.6, device=dev)
        with ops.options(vlad_parts=1):
            for _ in range(2):
                ops.vlad(toks, c)
            v = kernel_us(lambda: ops.vlad(toks, c), "vlad_fused")
        flat = toks.reshape(-1, 1536)
        with ops.options(kmeans_max_chunks=n_img):
            for _ in range(2):
                ops.kmeans_step(flat, c, "cosine", False)
            k = kernel_us(lambda: ops.kmeans_step(flat, c, "cosine", False), "kmeans_fused")
        tiles = (N + 15) // 16
        rows.append((n_img, N, tiles, v, k))
        print(f"{n_img:5d} images x {N:5d} tokens ({tiles:4d} tiles per workgroup): VLAD mode {v:8.1f} us, "
              f"k-means mode ({n_img} chunks) {k:8.1f} us", flush=True)
for n_img in (256, 1024):
    r = [x for x in rows if x[0] == n_img]
    if len(r) >= 2:
        for name, i, lo in (("VLAD mode", 3, 144), ("VLAD mode", 3, 1058), ("k-means mode", 4, 1058)):
            pts = [x for x in r if x[1] >= lo]
            if len(pts) < 2:
                continue
            (t0, t1) = (pts[0], pts[-1])
            slope = (t1[i] - t0[i]) / (t1[2] - t0[2])
            print(f"{n_img} images, {name}: {slope:.2f} us per tile, intercept {t0[i] - slope * t0[2]:.1f} us "
                  f"(from {t0[2]} and {t1[2]} tiles per workgroup)")
