"""One k-means step (anyloc_kmeans_step) at the config-4 size: ms per iteration and algorithmic TB/s.
usage: time_kmeans.py [all] [clustered]   -- "clustered": rows drawn around K modes (descriptor-like: top-2 cosine gaps of
~1e-1) instead of isotropic noise (gaps of ~1e-3: every eighth row goes through the exact re-scoring of the fused kernel)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops  # noqa: E402

dev = "cuda"
shapes = ((5_000_000, 1536, 32),) if "all" not in sys.argv else ((5_000_000, 1536, 32), (2_000_000, 1024, 32), (3_000_000, 384, 16))
clustered = "clustered" in sys.argv
for (rows, D, K) in shapes:
    if clustered:
        modes = torch.nn.functional.normalize(torch.randn(K, D, device=dev))
        x = modes[torch.randint(0, K, (rows,), device=dev)]
        x += 0.03 * torch.randn(rows, D, device=dev)
        x = torch.nn.functional.normalize(x)
        c = modes + 0.01 * torch.randn(K, D, device=dev)
    else:
        x = torch.nn.functional.normalize(torch.randn(rows, D, device=dev))
        c = x[torch.randperm(rows, device=dev)[:K]].clone()
    ops.kmeans_step(x, c, "cosine", True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops.kmeans_step(x, c, "cosine", True)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(json.dumps(dict(v=os.environ.get("ANYLOC_OPTIONS", "default"), data="clustered" if clustered else "isotropic", rows=rows, D=D, K=K, ms=round(ms, 3),
                          tb_s=round(rows * D * 4 / 1e9 / ms, 3))), flush=True)
    del x
