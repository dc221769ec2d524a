"""Stress of the fused VLAD launch for run-to-run reproducibility and agreement with the two-pass path: the same inputs REPS
times, bitwise compared with the first result; shapes with one and with several workgroups per image.  Used to bisect library
builds (tools/ab_libs/lib_<name>.so copied over anyloc_amd/libanyloc_hip.so by the job script).

    python tools/stress_vlad.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, synth  # noqa: E402

dev = "cuda"
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
total_bad = 0
for (n_img, N, D, K, parts) in ((300, 529, 1536, 32, 0), (1000, 529, 1536, 32, 0), (5, 529, 1536, 32, 0), (5, 529, 1536, 32, 8),
                                (7, 100, 768, 17, 0), (300, 300, 768, 32, 0), (3, 529, 768, 32, 0), (300, 257, 1024, 32, 0),
                                (300, 300, 384, 8, 0)):
    c = 0.8 * synth.clustered_tokens(1, K, D, n_modes=K, seed=3, device=dev)[0]
    toks = synth.clustered_tokens(n_img, N, D, n_modes=K, seed=11, noise=0.6, device=dev)
    with ops.options(vlad_two_pass=1):
        ref = ops.vlad(toks, c)
    with ops.options(vlad_parts=parts):
        first = ops.vlad(toks, c).clone()
        rel = ((first - ref).norm(dim=1) / ref.norm(dim=1))
        n_wrong = int((rel > 1e-5).sum())
        n_diff = 0
        for _ in range(REPS):
            again = ops.vlad(toks, c)
            n_diff += int((again != first).any(dim=1).sum())
    total_bad += n_wrong + n_diff
    print(f"n_img {n_img:5d} N {N:4d} D {D:4d} K {K:2d} parts {parts}: vs two-pass max rel {float(rel.max()):.2e} ({n_wrong} images > 1e-5); "
          f"{n_diff} image results differ from the first over {REPS} repeats", flush=True)
print("TOTAL BAD", total_bad)
