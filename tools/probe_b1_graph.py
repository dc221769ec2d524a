"""Does a hipGraph of the one-image forward shorten it?  (Dispatch timestamps put 98 % of the B = 1 forward INSIDE its ~225
kernels and 0.17 us between them -- tools/b1_gaps.py -- so the expectation is "no"; this measures it.)

    python tools/probe_b1_graph.py > gpurun_out/b1_graph.log
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, synth, weights  # noqa: E402

import utilities  # noqa: E402

name = "dinov2_vitg14"
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device="cuda", depth=40))
ext = utilities.DinoV2ExtractFeatures(name, 31, "value", device="cuda")
ext.dino_model.ffn_check = False                     # (the sampled telemetry read-back is a host sync: not capturable)
for hw in ((322, 322), (476, 630)):
    img = torch.randn(1, 3, *hw, device="cuda")
    for _ in range(5):
        ref = ext(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ext(img)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                ext(img)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = ext(img)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 50
        print(f"{hw[0]}x{hw[1]} one image: eager {eager * 1e3:.3f} ms, hipGraph replay {graph * 1e3:.3f} ms, "
              f"max |difference| {float((out - ref).abs().max()):.1e}", flush=True)
    except Exception as e:                              # noqa: BLE001
        print(f"{hw[0]}x{hw[1]} one image: eager {eager * 1e3:.3f} ms, capture failed: {type(e).__name__}: {str(e)[:300]}", flush=True)
