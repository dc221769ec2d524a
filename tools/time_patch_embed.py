"""Patch embedding on the two-term fp16 GEMM (option h3_patch = 1, default) against the fp32 matrix-core GEMM (0): ViT-G/14
322 x 322 forwards at B = 1 and B = 61, wall time per forward and the HIP-event times of the embedding's launches.

    python tools/time_patch_embed.py > gpurun_out/patch_embed.log
"""
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
ext.dino_model.ffn_check = False
for batch, n in ((1, 60), (61, 8)):
    img = torch.randn(batch, 3, 322, 322, generator=torch.Generator().manual_seed(1)).to(dev)
    toks = {}
    for rep in range(2):
        for mode in (0, 1):
            with ops.options(h3_patch=mode):
                for _ in range(3):
                    ext(img)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    tok = ext(img)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / n * 1e3
                ops.profile_enable(True)
                ops.profile_reset()
                for _ in range(4):
                    ext(img)
                torch.cuda.synchronize()
                dump = ops.profile_dump()
                prof = {t: dump[t]["ms"] / 4 for t in ("im2col", "split_h2", "vit_patch_embed_gemm", "cls_rows") if t in dump}
                ops.profile_enable(False)
            toks[mode] = tok.clone()
            print(f"B={batch} h3_patch={mode} rep {rep}: {wall:.3f} ms/forward  {batch / wall * 1e3:.1f} images/s  "
                  + "  ".join(f"{k} {v * 1e3:.1f} us" for k, v in prof.items()), flush=True)
    print(f"B={batch}: max |tokens(h3_patch=1) - tokens(0)| = {float((toks[1] - toks[0]).abs().max()):.2e} "
          f"(token scale {float(toks[0].abs().max()):.2f})", flush=True)
