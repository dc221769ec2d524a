"""Small-M plan sweep of the two-term fp16 block GEMMs (csrc/gemm_h3s.hip): ViT-G/14 322 x 322 forwards at B = 1, 2, 4 with
every (tile configuration, k-blocks per ring stage, split-K factor) forced through the h3s_* options; per GEMM kind the
time per launch from the library's HIP-event scopes, and the tokens' distance from the round-3 kernels (h3s_enable = 0).

    python tools/sweep_b1.py [batches, e.g. 1,2,4] [configurations, e.g. 2,4] [HxW, e.g. 476x630] [depth] > gpurun_out/b1_plan_sweep.log
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
H, W = (int(v) for v in sys.argv[3].split("x")) if len(sys.argv) > 3 else (322, 322)
DEPTH = int(sys.argv[4]) if len(sys.argv) > 4 else 32
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device=dev, depth=DEPTH))
ext = utilities.DinoV2ExtractFeatures(name, DEPTH - 1, "value", device=dev)
KSPLITS = (1, 2, 3, 4, 6, 8) if (H, W) != (322, 322) else (1, 2, 3, 4)
BATCHES = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (1, 2, 4)
TAGS = {"qkv": "vit_qkv_gemm", "proj": "vit_proj_gemm", "fc1": "vit_w12_gemm", "fc2": "vit_fc2_gemm"}
CFG_NAMES = ["64x64/2w", "64x128/2w(64x64)", "64x128/4w(32x64)", "64x128/2w(32x128)", "128x128/4w", "64x256/4w(64x64)",
             "64x256/4w(32x128)", "192x128/4w(96x64)"]
CFGS = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else tuple(range(len(CFG_NAMES)))


def run(img, n):
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
    per = {k: prof[t]["ms"] / prof[t]["calls"] * 1e3 for k, t in TAGS.items() if t in prof}    # us per launch
    other = {k: v["ms"] / n for k, v in prof.items() if k not in TAGS.values()}
    return wall, per, tok, other


for B in BATCHES:
    img = torch.randn(B, 3, H, W, device=dev)
    n = 6
    with ops.options(h3s_enable=0):
        wall0, per0, tok0, other0 = run(img, n)
    print(f"B={B} round-3 kernels: {wall0*1e3:.3f} ms/forward  " + "  ".join(f"{k}={v:.1f}us" for k, v in per0.items()), flush=True)
    print("      others (ms/forward): " + "  ".join(f"{k}={v:.3f}" for k, v in sorted(other0.items(), key=lambda kv: -kv[1])[:8]), flush=True)
    wall1, per1, tok1, _ = run(img, n)
    print(f"B={B} default plans:   {wall1*1e3:.3f} ms/forward  " + "  ".join(f"{k}={v:.1f}us" for k, v in per1.items()) +
          f"  max|dtok|={float((tok1 - tok0).abs().max()):.2e}", flush=True)
    best = {k: (per0[k], "round3") for k in per0}
    for cfg in CFGS:
        for kb in (1, 2, 4):
            for st in (3, 6):
                for ks in KSPLITS:
                    with ops.options(h3s_cfg=cfg, h3s_kb=kb, h3s_ksplit=ks, h3s_stages=st, h3s_mask=15):
                        try:
                            wall, per, tok, _ = run(img, 4)
                        except Exception as e:                                   # a plan the library rejects
                            print(f"  cfg={cfg} kb={kb} st={st} ks={ks}: {e}", flush=True)
                            continue
                    err = float((tok - tok0).abs().max())
                    print(f"  B={B} cfg={cfg}({CFG_NAMES[cfg]}) kb={kb} st={st} ks={ks}: {wall*1e3:.3f} ms  " +
                          "  ".join(f"{k}={v:.1f}" for k, v in per.items()) + f"  err={err:.1e}", flush=True)
                    if err < 5e-6:
                        for k, v in per.items():
                            if v < best[k][0]:
                                best[k] = (v, f"cfg={cfg} kb={kb} st={st} ks={ks}")
    print(f"B={B} BEST per GEMM: " + "  ".join(f"{k}: {v[0]:.1f}us [{v[1]}]" for k, v in best.items()), flush=True)
