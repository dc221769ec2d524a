"""Batched ViT-g forward (B = 61, 322 x 322: the headline step's extractor part): LayerNorm as lead workgroups interleaved with its
consumer GEMM's tiles (option h3_ln_lead = 1) against LayerNorm as launches of its own (= 0), interleaved A/B on one box.
Wall time per forward, per-kernel figures from the library's HIP-event scopes, bit equality of the tokens.
    python tools/probe_batched_ln_lead.py [B] > gpurun_out/batched_ln_lead.log"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, synth, weights  # noqa: E402

import utilities  # noqa: E402

dev = "cuda"
name = "dinov2_vitg14"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 61
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device=dev, depth=32))
ext = utilities.DinoV2ExtractFeatures(name, 31, "value", device=dev)
ext.dino_model.ffn_check = False
TAGS = {"qkv": "vit_qkv_gemm", "proj": "vit_proj_gemm", "w12": "vit_w12_gemm", "fc2": "vit_fc2_gemm", "attn": "attention",
        "ln": "layernorm_h2"}
img = torch.randn(B, 3, 322, 322, generator=torch.Generator().manual_seed(1)).to(dev)


def run(n=8):
    for _ in range(2):
        ext(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tok = ext(img)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    ops.profile_enable(True)
    ops.profile_reset()
    for _ in range(2):
        ext(img)
    torch.cuda.synchronize()
    prof = ops.profile_dump()
    ops.profile_enable(False)
    per = {k: prof[t]["ms"] / 2 for k, t in TAGS.items() if t in prof}
    return wall, per, tok.clone()


ref = None
VARIANTS = [dict(h3_ln_lead=0), dict(h3_ln_lead=1)]
for rep in range(int(os.environ.get("LEAD_REPS", "4"))):
    for kw in VARIANTS:
        with ops.options(**kw):
            w, p, t = run()
        if ref is None:
            ref = t
        print(f"B={B} {kw}: {w * 1e3:8.3f} ms/forward = {B / w:6.1f} images/s  per forward [ms]: " +
              "  ".join(f"{k}={v:6.2f}" for k, v in p.items()) + f"  sum={sum(p.values()):7.2f}  bits_equal={bool(torch.equal(t, ref))}", flush=True)
