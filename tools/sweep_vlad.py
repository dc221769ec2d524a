"""VLAD (K=32, 529 x 1536 tokens per image) and one k-means step: fused single-launch kernel vs the two-pass path.
usage: python tools/sweep_vlad.py [kmeans_rows]      env ANYLOC_OPTIONS=vlad_two_pass=1 / vlad_parts=n selects the path"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from anyloc_amd import ops, synth  # noqa: E402

dev = "cuda"
K, D, N = 32, 1536, 529
centers = (0.8 * synth.clustered_tokens(1, K, D, n_modes=K, seed=1)[0]).to(dev)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


mode = "options[" + os.environ.get("ANYLOC_OPTIONS", "") + "]"
for n_img in (1, 4, 16, 61, 122, 256, 1024):
    x = synth.clustered_tokens(n_img, N, D, n_modes=K, seed=3, device=dev)
    ms = timeit(lambda: ops.vlad(x, centers))
    gb = n_img * (N * D + 2 * K * D) * 4 / 1e9
    print(json.dumps(dict(what="vlad", mode=mode, images=n_img, ms=round(ms, 4), tb_s=round(gb / ms, 3))), flush=True)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
x = torch.nn.functional.normalize(torch.randn(rows, D, device=dev))
ms = timeit(lambda: ops.kmeans_step(x, centers, "cosine", True), n=5)
print(json.dumps(dict(what="kmeans_step", mode=mode, rows=rows, ms=round(ms, 3), tb_s=round(rows * D * 4 / 1e9 / ms, 3))), flush=True)
