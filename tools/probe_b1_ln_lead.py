"""One image per call, ViT-g: LayerNorm as the LEAD role of its consumer GEMM's launch (option h3s_ln_lead = 1: LN1 + qkv and
LN2 + w12 are one launch each, 5 launches per block instead of 7) against LayerNorm as a launch of its own (= 0), interleaved.
Wall time per forward, per-launch figures from the library's HIP-event scopes, bit equality of the tokens, at 322 x 322
(530 rows) and 476 x 630 (the scripts' default, 1 531 rows).
    python tools/probe_b1_ln_lead.py > gpurun_out/b1_ln_lead.log"""
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
TAGS = {"qkv": "vit_qkv_gemm", "proj": "vit_proj_gemm", "w12": "vit_w12_gemm", "fc2": "vit_fc2_gemm", "attn": "attention",
        "ln": "layernorm_h2"}


def run(img, n=40):
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
    calls = sum(v["calls"] for v in prof.values()) // 6
    return wall, per, calls, tok.clone()


for hw in ((322, 322), (476, 630), (224, 224)):
    img = torch.randn(1, 3, *hw, generator=torch.Generator().manual_seed(hw[0])).to(dev)
    for check in (True, False):
        ext.dino_model.ffn_check = check
        ref = None
        for rep in range(3):
            for lead in (0, 1):
                with ops.options(h3s_ln_lead=lead):
                    w, p, calls, t = run(img)
                if ref is None:
                    ref = t
                print(f"{hw[0]}x{hw[1]} ffn_check={int(check)} h3s_ln_lead={lead}: {w * 1e3:7.3f} ms/forward  launches={calls}  " +
                      "  ".join(f"{k}={v:5.1f}us" for k, v in p.items()) + f"  bits_equal={bool(torch.equal(t, ref))}", flush=True)
