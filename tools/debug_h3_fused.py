"""Bisect the fused two-term fp16 forward: tokens of a short ViT in ANYLOC_H3_FUSE = 0 (fp32 round trips), 2 (attention
side fused), 3 (FFN side fused), 1 (both) against the fp32-MFMA forward.  usage: python tools/debug_h3_fused.py [model] [depth] [B] [hw]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import synth, weights  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dinov2_vitg14"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
hw = int(sys.argv[4]) if len(sys.argv) > 4 else 322
sd = synth.synthetic_state_dict(name, 0, device="cuda", depth=depth)
weights.register_state_dict(name, sd)
import utilities  # noqa: E402
img = torch.randn(B, 3, hw, hw, device="cuda")
outs = {}
for tag, gemm, fuse in (("f32", "f32", "1"), ("h3 unfused", "h3", "0"), ("h3 attn", "h3", "2"), ("h3 ffn", "h3", "3"), ("h3 fused", "h3", "1")):
    os.environ["ANYLOC_GEMM"], os.environ["ANYLOC_H3_FUSE"] = gemm, fuse
    ext = utilities.DinoV2ExtractFeatures(name, depth - 1, "token", device="cuda")
    outs[tag] = ext(img)
    torch.cuda.synchronize()
    d = (outs[tag] - outs["f32"]).abs()
    print(f"{tag:12s} max |tok - f32| = {float(d.max()):.3e}   finite={bool(torch.isfinite(outs[tag]).all())}", flush=True)
