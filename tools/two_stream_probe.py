"""Do two half-batches on two HIP streams beat one batch on one stream?  The halves are independent (own workspaces), so the
GPU may fill one half's LayerNorm / attention / GEMM tails with the other half's matrix work.
    python tools/two_stream_probe.py > gpurun_out/two_stream_probe.log"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import synth, weights  # noqa: E402

import utilities  # noqa: E402

dev = "cuda"
name = "dinov2_vitg14"
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device=dev, depth=32))
ext = utilities.DinoV2ExtractFeatures(name, 31, "value", device=dev)
ext.dino_model.ffn_check = False
streams = [torch.cuda.Stream() for _ in range(4)]


def run_split(img, parts):
    if len(parts) == 1:
        return ext(img)
    outs, s0, cur = [], 0, torch.cuda.current_stream()
    for st, n in zip(streams, parts):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(ext(img[s0:s0 + n]))
        s0 += n
    for st in streams[:len(parts)]:
        cur.wait_stream(st)
    return torch.cat(outs)


for B, splits in ((61, [[61], [30, 31], [31, 30], [20, 20, 21], [15, 15, 15, 16]]), (122, [[122], [61, 61]])):
    img = torch.randn(B, 3, 322, 322, device=dev)
    ref = None
    for parts in splits:
        for _ in range(2):
            tok = run_split(img, parts)
        torch.cuda.synchronize()
        n = 8
        t0 = time.perf_counter()
        for _ in range(n):
            tok = run_split(img, parts)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        if ref is None:
            ref = tok
        print(f"B={B} parts={parts}: {dt*1e3:.2f} ms  {B/dt:.1f} images/s  max|dtok| vs one stream {float((tok - ref).abs().max()):.1e}", flush=True)
