"""One-image forwards with the projection / fc2 GEMMs leaving split-K slabs that the next LayerNorm reduces (option
h3s_consumer, csrc/vit.hip): wall time per forward and per-launch times of proj, fc2 and LayerNorm against the in-GEMM
epilogues, for forced (tile configuration, k-blocks per stage, ring depth, split factor) of those two GEMMs.

    python tools/sweep_b1_consumer.py [batches] > gpurun_out/b1_consumer_splitk.log"""
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
ext.dino_model.ffn_check_every = 0
BATCHES = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 2)
TAGS = {"proj": "vit_proj_gemm", "fc2": "vit_fc2_gemm", "ln": "layernorm_h2", "qkv": "vit_qkv_gemm", "fc1": "vit_w12_gemm",
        "attn": "attention"}


def run(img, n=6):
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
    for _ in range(n):
        ext(img)
    torch.cuda.synchronize()
    prof = ops.profile_dump()
    ops.profile_enable(False)
    per = {k: prof[t]["ms"] / prof[t]["calls"] * 1e3 for k, t in TAGS.items() if t in prof}
    return wall, per, tok


def line(tag, wall, per, err):
    return f"{tag}: {wall*1e3:.3f} ms/forward  " + "  ".join(f"{k}={v:.1f}" for k, v in per.items()) + f"  err={err:.1e}"


for B in BATCHES:
    img = torch.randn(B, 3, 322, 322, device=dev)
    with ops.options(h3s_consumer=0):
        w0, p0, t0_ = run(img)
    print(line(f"B={B} in-GEMM epilogues (h3s_consumer=0)", w0, p0, 0.0), flush=True)
    w1, p1, t1 = run(img)
    print(line(f"B={B} slabs + LayerNorm reduce, default plans", w1, p1, float((t1 - t0_).abs().max())), flush=True)
    best = (w1, "default")
    for cfg in (0, 2, 1, 4):
        for kb, st in ((1, 3), (1, 6), (2, 3), (2, 6)):
            for ks in (1, 2, 3, 4, 6, 8):
                with ops.options(h3s_cfg=cfg, h3s_kb=kb, h3s_stages=st, h3s_ksplit=ks, h3s_mask=10):
                    w, p, t = run(img, 4)
                err = float((t - t0_).abs().max())
                print(line(f"  B={B} cfg={cfg} kb={kb} st={st} ks={ks}", w, p, err), flush=True)
                if err < 5e-6 and w < best[0]:
                    best = (w, f"cfg={cfg} kb={kb} st={st} ks={ks}")
    print(f"B={B} BEST forward: {best[0]*1e3:.3f} ms [{best[1]}]", flush=True)
