"""CPU numerics study for a split-bf16 ("bf16x6") GEMM path: every fp32 operand is split exactly into three
bf16 terms (a = a1 + a2 + a3) and the product is rebuilt from six bf16 x bf16 -> fp32 MFMA-style products
(a1b1, a1b2, a2b1, a1b3, a3b1, a2b2).  On MI355X that would run on v_mfma_f32_32x32x16_bf16 at 16x the
fp32-MFMA rate, i.e. 2.67x the fp32 roofline for the six products.  This script measures, on the CPU and with the
oracle DINOv2 (config 1: ViT-S/14 layer 9 'value', K=8), how far tokens / cluster ids / VLADs move when ALL
linear layers (and the patch-embed) use the emulated split product instead of fp32 -- the evidence asked for
before any such kernel may replace the exact-fp32 one (north_star parity bar: ids identical, VLAD <= 1e-5)."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import synth  # noqa: E402
from oracle import dinov2_ref, vlad_ref  # noqa: E402

T6 = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
T3 = [(0, 0), (0, 1), (1, 0)]


def split3(x):
    x1 = x.bfloat16().float()
    r = x - x1
    x2 = r.bfloat16().float()
    x3 = (r - x2).bfloat16().float()
    return x1, x2, x3


def make_linear(terms):
    real = F.linear

    def linear(x, w, b=None):
        xs, ws = split3(x), split3(w)
        out = None
        for i, j in terms:
            t = real(xs[i], ws[j])
            out = t if out is None else out + t
        return out if b is None else out + b
    return linear


def run(terms):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "config1_vits14_l9_value_k8.npz"))
    sd = synth.synthetic_state_dict("dinov2_vits14", 0)
    model = dinov2_ref.build("dinov2_vits14", sd)
    db, qu, gt = synth.synthetic_places(24, 8, 224, 224, seed=42)
    imgs = torch.cat([db, qu])
    ref_tok = torch.cat([dinov2_ref.extract_facet(model, im[None], 9, "value") for im in imgs])
    real = F.linear
    F.linear = make_linear(terms)
    torch.nn.functional.linear = F.linear
    try:
        tok = torch.cat([dinov2_ref.extract_facet(model, im[None], 9, "value") for im in imgs])
    finally:
        F.linear = real
        torch.nn.functional.linear = real
    centers = torch.from_numpy(g["centers"])
    lab_ref = torch.stack([vlad_ref.hard_labels(t, centers) for t in ref_tok])
    lab = torch.stack([vlad_ref.hard_labels(t, centers) for t in tok])
    v_ref = torch.stack([vlad_ref.vlad_hard(t, centers)[0] for t in ref_tok])
    v = torch.stack([vlad_ref.vlad_hard(t, centers)[0] for t in tok])
    clean = (lab == lab_ref).all(dim=1)
    rel = (v - v_ref).norm(dim=1) / v_ref.norm(dim=1)
    return dict(token_max_abs=float((tok - ref_tok).abs().max()), label_flips=int((lab != lab_ref).sum()),
                labels=int(lab.numel()), vlad_rel_max_clean=float(rel[clean].max()) if clean.any() else None)


if __name__ == "__main__":
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    print("bf16x6:", run(T6))
    print("bf16x3:", run(T3))
