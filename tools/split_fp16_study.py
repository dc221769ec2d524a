"""CPU numerics study for a row-scaled two-term fp16 split ("h3") GEMM: every operand row is scaled by a power of
two so that its largest magnitude lies in [2^14, 2^15), x*2^e = h + l with h = fp16(x*2^e), l = fp16(x*2^e - h)
(22 mantissa bits; l needs no extra scaling because the row scale keeps it a normal fp16 for every element within
2^16 of the row maximum), and the product is rebuilt from THREE fp16 x fp16 -> fp32 products (h h, h l, l h) in one
fp32 accumulator, then descaled by 2^-(e_row + e_col).  On MI355X that would be three v_mfma_f32_32x32x16_f16 per
k-step instead of the six bf16 ones of gemm_x6.hip -- half the matrix-core passes of a kernel that is power-limited.

Measured here with the oracle DINOv2 (config 1) like tools/split_bf16_study.py: token / cluster-id / VLAD deviations
when ALL linear layers use the emulated product, plus a stress variant with injected activation outliers (x1000 on
0.1 % of the entries of every linear input), and the plain-GEMM error against float64."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import synth  # noqa: E402
from oracle import dinov2_ref, vlad_ref  # noqa: E402


def split_h2(x):
    """rows of x (last dim = k) -> (h, l, scale) with x ~= (h + l) / scale."""
    amax = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    e = 14 - torch.floor(torch.log2(amax))                   # amax * 2^e in [2^14, 2^15)
    scale = torch.exp2(e)
    xs = x * scale                                           # exact (power of two)
    h = xs.half().float()
    l = (xs - h).half().float()
    assert torch.isfinite(h).all()
    return h, l, scale


def linear_h3(real):
    def linear(x, w, b=None):
        xh, xl, sx = split_h2(x)
        wh, wl, sw = split_h2(w)
        acc = real(xh, wh) + (real(xh, wl) + real(xl, wh))
        out = acc / sx / sw.reshape(-1)
        return out if b is None else out + b
    return linear


def outlier_wrap(lin, gen):
    def linear(x, w, b=None):
        m = torch.rand(x.shape, generator=gen) < 1e-3
        return lin(torch.where(m, x * 1000.0, x), w, b)
    return linear


def run(make, stress=False):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "config1_vits14_l9_value_k8.npz"))
    sd = synth.synthetic_state_dict("dinov2_vits14", 0)
    model = dinov2_ref.build("dinov2_vits14", sd)
    db, qu, gt = synth.synthetic_places(24, 8, 224, 224, seed=42)
    imgs = torch.cat([db, qu])
    real = F.linear

    def tokens(lin):
        F.linear = lin
        torch.nn.functional.linear = lin
        try:
            return torch.cat([dinov2_ref.extract_facet(model, im[None], 9, "value") for im in imgs])
        finally:
            F.linear = real
            torch.nn.functional.linear = real
    if stress:       # the SAME outlier pattern in the reference and in the emulated run
        ref_tok = tokens(outlier_wrap(real, torch.Generator().manual_seed(1)))
        tok = tokens(outlier_wrap(make(real), torch.Generator().manual_seed(1)))
    else:
        ref_tok, tok = tokens(real), tokens(make(real))
    centers = torch.from_numpy(g["centers"])
    lab_ref = torch.stack([vlad_ref.hard_labels(t, centers) for t in ref_tok])
    lab = torch.stack([vlad_ref.hard_labels(t, centers) for t in tok])
    v_ref = torch.stack([vlad_ref.vlad_hard(t, centers)[0] for t in ref_tok])
    v = torch.stack([vlad_ref.vlad_hard(t, centers)[0] for t in tok])
    clean = (lab == lab_ref).all(dim=1)
    rel = (v - v_ref).norm(dim=1) / v_ref.norm(dim=1)
    return dict(token_max_abs=float((tok - ref_tok).abs().max()), label_flips=int((lab != lab_ref).sum()),
                labels=int(lab.numel()), vlad_rel_max_clean=float(rel[clean].max()) if clean.any() else None)


def gemm_probe():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(512, 1536, generator=g) * (0.25 + torch.rand(512, 1, generator=g))
    a[:, ::97] *= 300.0                                        # heavy-tailed columns inside every row
    w = torch.randn(768, 1536, generator=g) * 0.02
    ref = a.double() @ w.double().t()
    mag = a.double().abs() @ w.double().abs().t()
    out = {}
    out["fp32"] = float(((a @ w.t()).double() - ref).abs().div(mag).max())
    out["h3"] = float((linear_h3(F.linear)(a, w).double() - ref).abs().div(mag).max())
    return out


if __name__ == "__main__":
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    print("plain GEMM max |err| / sum|a||b| :", gemm_probe())
    print("h3 (all linear layers):", run(linear_h3))
    print("h3, outlier stress     :", run(linear_h3, stress=True))
