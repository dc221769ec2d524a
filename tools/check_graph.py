"""HIP-graph replay of small-batch forwards (ANYLOC_VIT_GRAPH, csrc/vit.hip): bit-equality with the eager launch
sequence on changing inputs, and ms per call with / without (loop with a .cpu() per call, as the reference's scripts do,
and back-to-back)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import synth, weights  # noqa: E402

import utilities  # noqa: E402

dev = "cuda"
name = sys.argv[1] if len(sys.argv) > 1 else "dinov2_vitg14"
layer = {"dinov2_vitg14": 31, "dinov2_vitl14": 23, "dinov2_vitb14": 11, "dinov2_vits14": 9}[name]
weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, device=dev, depth=layer + 1))
ext = utilities.DinoV2ExtractFeatures(name, layer, "value", device=dev)
g = torch.Generator(device=dev)
g.manual_seed(1)
for B in (1, 2, 4):
    imgs = [torch.randn(B, 3, 322, 322, generator=g, device=dev) for _ in range(4)]
    os.environ["ANYLOC_VIT_GRAPH_MAX_ROWS"] = "0"
    ref = [ext(im).clone() for im in imgs]
    res = {}
    for label, rows in (("eager", "0"), ("graph", "2200")):
        os.environ["ANYLOC_VIT_GRAPH_MAX_ROWS"] = rows
        outs = [ext(im) for im in imgs + imgs]                      # eager, capture, replays
        same = all(torch.equal(o, r) for o, r in zip(outs, ref + ref))
        torch.cuda.synchronize()
        n = 40
        t0 = time.perf_counter()
        for i in range(n):
            ext(imgs[i & 3])
        torch.cuda.synchronize()
        back2back = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for i in range(n):
            ext(imgs[i & 3]).cpu()
        percall = (time.perf_counter() - t0) / n
        res[label] = (back2back, percall, same)
    e, gr = res["eager"], res["graph"]
    print(f"{name} B={B}: eager {e[0]*1e3:.2f} ms back-to-back / {e[1]*1e3:.2f} ms with .cpu();  graph {gr[0]*1e3:.2f} / "
          f"{gr[1]*1e3:.2f} ms;  bit-equal eager={e[2]} graph={gr[2]}  (graphs, replays)={ext.dino_model.graph_stats()}", flush=True)
