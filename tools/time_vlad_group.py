"""Fused VLAD kernel with one centre request per token against label-grouped requests (option vlad_group: a token with its
predecessor's label reuses that token's centre columns), on tokens whose cluster membership is random per token (the bench
stage's worst case) and spatially coherent (runs of 4 / 16 equal labels, like raster-ordered patch tokens).
    python tools/time_vlad_group.py > gpurun_out/vlad_group.log"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops  # noqa: E402

dev = "cuda"
D, K, N = 1536, 32, 529
g = torch.Generator(device=dev)
g.manual_seed(3)
modes = torch.nn.functional.normalize(torch.randn(K, D, generator=g, device=dev), dim=1)
centers = 0.7 * modes
for n_img in (256, 61):
    for run in (1, 4, 16):
        pick = torch.randint(0, K, (n_img, (N + run - 1) // run), generator=g, device=dev).repeat_interleave(run, dim=1)[:, :N]
        x = torch.nn.functional.normalize(modes[pick] + (0.6 / math.sqrt(D)) * torch.randn(n_img, N, D, generator=g, device=dev), dim=-1)
        out = {}
        for grp in (0, 1):
            with ops.options(vlad_group=grp):
                for _ in range(3):
                    v = ops.vlad(x, centers)
                torch.cuda.synchronize()
                ops.profile_enable(True)
                ops.profile_reset()
                for _ in range(20):
                    v = ops.vlad(x, centers)
                torch.cuda.synchronize()
                ops.profile_enable(False)
                ms = ops.profile_dump()["vlad_fused"]["ms"] / 20
            out[grp] = (ms, v)
        same = bool(torch.equal(out[0][1], out[1][1]))
        gb = n_img * N * D * 4 / 1e9
        print(f"{n_img:4d} images, label runs of {run:2d}: per-token requests {out[0][0]:.4f} ms ({gb / out[0][0]:.2f} TB/s)   "
              f"label-grouped {out[1][0]:.4f} ms ({gb / out[1][0]:.2f} TB/s)   same bits: {same}", flush=True)
