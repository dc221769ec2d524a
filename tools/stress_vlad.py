"""Stress of the fused VLAD launch for run-to-run reproducibility and agreement with the two-pass path: the same inputs REPS
times, bitwise compared with the first result; shapes with one and with several workgroups per image.

    python tools/stress_vlad.py [reps] [--variants 0,1,2,...]

Round 5 bisected library builds with it (tools/ab_libs/lib_<name>.so copied over anyloc_amd/libanyloc_hip.so by the job script).
Round 6: the variants of the register-indexed gather are compiled into ONE library (option ``vlad_gather_v``, D = 1536 VLAD mode:
csrc/vlad_fused.hip, ``fused3_kernel<..., GV>``); ``--variants`` runs the D = 1536 shapes once per variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops, synth  # noqa: E402

dev = "cuda"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
REPS = int(args[0]) if args else 20
variants = [0]
for i, a in enumerate(sys.argv):
    if a == "--variants":
        variants = [int(v) for v in sys.argv[i + 1].split(",")]
    elif a.startswith("--variants="):
        variants = [int(v) for v in a.split("=", 1)[1].split(",")]

SHAPES = ((300, 529, 1536, 32, 0), (1000, 529, 1536, 32, 0), (5, 529, 1536, 32, 0), (5, 529, 1536, 32, 8),
          (7, 100, 768, 17, 0), (300, 300, 768, 32, 0), (3, 529, 768, 32, 0), (300, 257, 1024, 32, 0),
          (300, 300, 384, 8, 0))
grand = {}
for gv in variants:
    total_bad = 0
    print(f"=== vlad_gather_v = {gv}", flush=True)
    for (n_img, N, D, K, parts) in SHAPES:
        if gv != 0 and D != 1536:
            continue                      # the variants exist for the D = 1536 instantiation only
        c = 0.8 * synth.clustered_tokens(1, K, D, n_modes=K, seed=3, device=dev)[0]
        toks = synth.clustered_tokens(n_img, N, D, n_modes=K, seed=11, noise=0.6, device=dev)
        with ops.options(vlad_two_pass=1):
            ref = ops.vlad(toks, c)
        with ops.options(vlad_parts=parts, vlad_gather_v=gv):
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
    print("TOTAL BAD", total_bad, flush=True)
    grand[gv] = total_bad
if len(variants) > 1:
    print("SUMMARY (variant: bad results)", grand)
