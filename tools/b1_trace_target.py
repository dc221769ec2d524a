"""Target of `rocprofv3 --kernel-trace` for the one-image-per-call regime: ViT-g/14, 322 x 322, layer 31
'value', B = 1, N forwards back to back (one synchronise per forward, as `.cpu()` in the reference scripts forces), no
HIP-event brackets.  Prints the wall time per forward; tools/b1_gaps.py turns the kernel trace of this run into the
split "inside kernels / between kernels".
    python tools/b1_trace_target.py [forwards=12]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from anyloc_amd import _lib, synth, weights  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    torch.cuda.set_device(0)
    _lib.load()
    import utilities
    name = "dinov2_vitg14"
    weights.register_state_dict(name, synth.synthetic_state_dict(name, seed=0, device="cuda:0"))
    ext = utilities.DinoV2ExtractFeatures(name, 31, "value", device="cuda:0")
    _, qu, _ = synth.synthetic_places(8, 8, 322, 322, seed=42, device="cuda:0")
    for i in range(4):                                   # warm-up: telemetry pass, workspaces, clocks
        ext(qu[i:i + 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        ext(qu[i % 8:i % 8 + 1])
        torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    print(json.dumps({"forwards": n, "wall_ms_per_forward": round(el * 1e3, 3), "warmup_forwards": 4}), flush=True)


if __name__ == "__main__":
    main()
