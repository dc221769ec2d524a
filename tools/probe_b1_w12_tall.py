"""One-image ViT-g forward: the FFN input GEMM (w12, N = 8192) on 192 x 128 tiles (3 x 64 = 192 equal workgroups, one per
CU; option h3s_w12_tall = 1) against 128 x 128 tiles (320 workgroups, two on 64 of the CUs; = 0), interleaved; plus the
192-row tile forced on the other block GEMMs for the record.  Wall time per forward, time per launch from the library's
HIP-event scopes, distance of the tokens from the 128 x 128 plan and from the same image inside a batch.
    python tools/probe_b1_w12_tall.py > gpurun_out/b1_w12_tall.log"""
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
TAGS = {"qkv": "vit_qkv_gemm", "proj": "vit_proj_gemm", "w12": "vit_w12_gemm", "fc2": "vit_fc2_gemm"}
_, qu, _ = synth.synthetic_places(8, 8, 322, 322, seed=42, device=dev)
img = qu[3:4]


def run(n=30):
    for _ in range(3):
        ext(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tok = ext(img)
        torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    ops.profile_enable(True)
    ops.profile_reset()
    for _ in range(6):
        ext(img)
    torch.cuda.synchronize()
    prof = ops.profile_dump()
    ops.profile_enable(False)
    per = {k: prof[t]["ms"] / prof[t]["calls"] * 1e3 for k, t in TAGS.items() if t in prof}
    return wall, per, tok.clone()


def show(tag, wall, per, tok, ref):
    print(f"{tag:<44s} {wall * 1e3:7.3f} ms/forward  " + "  ".join(f"{k}={v:5.1f}us" for k, v in per.items()) +
          f"  max|dtok|={float((tok - ref).abs().max()):.2e}", flush=True)


batch = ext(qu)
with ops.options(h3s_w12_tall=0):
    w0, p0, t0 = run()
show("w12 128x128 (h3s_w12_tall=0)", w0, p0, t0, t0)
for rep in range(2):
    w1, p1, t1 = run()
    show("w12 192x128 (default)", w1, p1, t1, t0)
    with ops.options(h3s_w12_tall=0):
        w, p, t = run()
    show("w12 128x128 (h3s_w12_tall=0)", w, p, t, t0)
print(f"default plan vs the same image inside a batch of 8 (position 3): max|dtok| = {float((t1 - batch[3:4]).abs().max()):.2e}", flush=True)
for tag, kw in (("w12 192x128, two k-blocks per stage", dict(h3s_cfg=7, h3s_kb=2, h3s_mask=4)),
                ("w12 192x128, 6-deep ring", dict(h3s_cfg=7, h3s_stages=6, h3s_mask=4)),
                ("w12 192x128, split-K 2", dict(h3s_cfg=7, h3s_ksplit=2, h3s_mask=4)),
                ("qkv 192x128 (108 workgroups)", dict(h3s_cfg=7, h3s_mask=1)),
                ("qkv 192x128, split-K 2 (216)", dict(h3s_cfg=7, h3s_ksplit=2, h3s_mask=1)),
                ("fc2 192x128, split-K 6 (216)", dict(h3s_cfg=7, h3s_ksplit=6, h3s_mask=8)),
                ("fc2 128x128, split-K 4 (240)", dict(h3s_cfg=4, h3s_ksplit=4, h3s_mask=8)),
                ("proj 192x128, split-K 6 (216)", dict(h3s_cfg=7, h3s_ksplit=6, h3s_mask=2))):
    with ops.options(**kw):
        w, p, t = run(12)
    show(tag, w, p, t, t0)
