"""Sequence lengths the reference's scripts really produce, against the CPU oracle (GPU box only).

The bench and the full-size parity tests sit at 322 x 322 (T = 530) and 518 x 518 (T = 1370).  The reference's own
drivers feed other shapes through the same extractor, one image per call:
  * ``configs.py:141`` ``resize = [480, 640]`` -> ``scripts/dino_v2_vlad.py:173-176`` centre-crops to multiples of 14:
    476 x 630 = 34 x 45 patches, **T = 1531**;
  * VPAir in ``scripts/dino_v2_vlad.py:168`` is 2 394 patches (588 x 798 = 42 x 57), **T = 2395**;
  * ``demo/anyloc_vlad_generate.py:56,165-181`` admits images up to 1 024 px on the longer side: 1 022 x 1 022 =
    73 x 73 patches, **T = 5330**.
T > 1 984 puts an image on more than 64 key groups of 32 rows: ``attention_h3_kernel`` then reads the per-tile K / V
scales from memory instead of a lane register (``csrc/attention.hip``, "longer images fall back to loads"); the
positional table is interpolated to non-square and larger-than-native grids.  Every test compares the HIP path
(through the C ABI) with the CPU oracle -- float64 for the attention kernels, the restated hub model
(``oracle/dinov2_ref.py``, reference ``utilities.py:263-285``) for the extractor -- at the bars of the fixed-shape
tests: attention <= 2e-5 abs, unit-norm tokens <= 2e-5 max-abs in all three GEMM arithmetics, B = 1 and B = 3.
"""
import pytest
import torch

from anyloc_amd import synth, weights
from oracle import dinov2_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOKEN_ATOL = 2e-5
# (H, W) in pixels -> tokens incl. CLS
SCRIPT_DEFAULT = (476, 630)      # T = 1531
VPAIR = (588, 798)               # T = 2395
DEMO_CAP = (1022, 1022)          # T = 5330


def _attention_f64(qkv, heads, want_smax=False):
    """softmax((q / 8) k^T) v in float64, one head at a time (T = 5330: 227 MB per head); ``want_smax``: also the
    largest |logit| of every query row over its heads."""
    B, T, D3 = qkv.shape
    D = D3 // 3
    out = torch.empty(B, T, D, dtype=torch.float64)
    smax = torch.zeros(B, T, dtype=torch.float64)
    x = qkv.double().reshape(B, T, 3, heads, 64)
    for b in range(B):
        for h in range(heads):
            q, k, v = x[b, :, 0, h], x[b, :, 1, h], x[b, :, 2, h]
            s = (q * 0.125) @ k.t()
            smax[b] = torch.maximum(smax[b], s.abs().amax(dim=1))
            out[b, :, h * 64:(h + 1) * 64] = torch.softmax(s, dim=-1) @ v
    return (out, smax) if want_smax else out


def _spiky_qkv(B, T, heads, seed):
    D = heads * 64
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn(B, T, 3 * D, generator=g) * 1.5
    qkv[0, 3, :D] *= 6.0                                  # a spiky query / key pair: the running-max rescale path
    qkv[0, T - 2, D:2 * D] *= 6.0
    qkv[:, ::7] *= 0.05                                   # small-magnitude tokens next to ordinary ones
    qkv[:, 5, 2 * D:] *= 40.0                             # one value row far above the others (sets the image scale)
    if T > 2080:
        qkv[0, 2048:2080, 2 * D:] = 0.0                   # an all-zero V tile beyond key group 64: must not set the scale
    qkv[B - 1, T - 40:, D:2 * D] *= 3.0                   # loud keys in the LAST key groups of the last image
    return qkv


LONG_T = [(1, 1531, 24), (3, 1531, 2), (2, 2395, 3), (1, 5330, 2), (3, 5330, 1), (2, 2047, 2), (2, 2049, 1)]


@pytest.mark.parametrize("ks", [1, 2], ids=["keys-unsplit", "keys-split-2"])
@pytest.mark.parametrize("B,T,heads", LONG_T)
def test_attention_h3_long_sequences(B, T, heads, ks):
    """``anyloc_attention_h3`` (the default forward's attention) at the script / VPAir / demo-cap lengths and on both
    sides of the 64-key-group boundary (T = 2047: image 0 lies on 64 global 32-row groups -- scales in a lane register --
    and image 1, starting inside group 63, on 65 -- scales from memory -- in ONE launch; 2049: always 65), images sharing
    32-row groups (T % 32 != 0), against float64."""
    from anyloc_amd import _lib, ops
    _lib.load()
    ops.set_option("attn_h3_ks", ks)                      # both workgroup shapes (the library picks by grid size)
    D = heads * 64
    qkv = _spiky_qkv(B, T, heads, B * T + heads)
    img, inv = ops.attention_h3(qkv.to(DEV), heads)
    out = ops.h2_image_to_f32(img, inv, B * T, D).reshape(B, T, D).cpu()
    ref = _attention_f64(qkv, heads)
    vmax = qkv[:, :, 2 * D:].abs().amax(dim=(1, 2)).double()
    err = (out - ref).abs().amax(dim=(1, 2))
    # every row of an image carries 22 bits relative to the image's largest |v| (tests/test_gpu_kernels.py)
    assert bool((err <= 3e-6 * vmax + 2e-6).all()), (err, vmax)
    assert bool(torch.isfinite(out).all())
    rel = float((out.double() - ref).abs().max() / ref.abs().max())
    assert rel < 5e-6, rel
    inv_c = inv.cpu().reshape(B, T)
    assert bool((inv_c == inv_c[:, :1]).all())            # one power-of-two scale per image


@pytest.mark.parametrize("kernel", ["fp32-mfma", "split-bf16"])
@pytest.mark.parametrize("B,T,heads", [(1, 1531, 6), (2, 2395, 2), (1, 5330, 2), (3, 5330, 1)])
def test_attention_long_sequences(B, T, heads, kernel):
    """The other two attention kernels (``anyloc_attention``: exact-fp32 MFMA, and the six-product split-bf16 one of the
    x6 forward) at the same lengths, against float64."""
    from anyloc_amd import _lib, ops
    _lib.load()
    ops.set_option("attn_x6", 1 if kernel == "split-bf16" else 0)
    D = heads * 64
    g = torch.Generator().manual_seed(B * T + heads)
    qkv = torch.randn(B, T, 3 * D, generator=g) * 1.5
    qkv[0, 3, :D] *= 6.0
    qkv[0, T - 2, D:2 * D] *= 6.0
    out = ops.attention(qkv.to(DEV), heads).cpu()
    ref, smax = _attention_f64(qkv, heads, want_smax=True)
    # fp32 logits carry 2^-24 |s| of rounding, i.e. that RELATIVE error on the row's probabilities (the reference's fp32
    # attention has the same): the rows of the spiky query / key reach |s| ~ 200 here, every other row sits at the 2e-5 bar
    # of the fixed-shape tests
    vmax = float(qkv[:, :, 2 * D:].abs().max())
    err = (out.double() - ref).abs().amax(dim=2)
    bar = 2e-5 + 2.0 ** -22 * smax * vmax
    assert bool((err <= bar).all()), (float(err.max()), float((err / bar).max()))
    assert float(err.max() / ref.abs().max()) < 1e-5


class _Model:
    """Oracle side of one (architecture, depth): weights + the restated hub model, built once per module."""

    def __init__(self, name, depth, layer, seed):
        self.name, self.depth, self.layer = name, depth, layer
        self.sd = synth.synthetic_state_dict(name, seed, depth=depth)
        m = dinov2_ref.DinoVisionTransformer(name)
        m.blocks = m.blocks[:depth]
        m.load_state_dict(self.sd, strict=True)
        self.model = m.eval()
        self._ref = {}

    def ref_tokens(self, hw, n_img, facet):
        key = (hw, n_img, facet)
        if key not in self._ref:
            g = torch.Generator().manual_seed(hw[0] * 7 + hw[1])
            # image-like input: smooth structure + noise, ImageNet-normalised range
            imgs = torch.randn(n_img, 3, hw[0] // 14, hw[1] // 14, generator=g)
            imgs = torch.nn.functional.interpolate(imgs, size=hw, mode="bilinear", align_corners=False)
            imgs = imgs + 0.3 * torch.randn(n_img, 3, *hw, generator=g)
            with torch.no_grad():                         # B = 1 per call: the reference's calling convention
                toks = torch.cat([dinov2_ref.extract_facet(self.model, im[None], self.layer, facet) for im in imgs])
            self._ref[key] = (imgs, toks)
        return self._ref[key]


@pytest.fixture(scope="module")
def vits_4blocks():
    return _Model("dinov2_vits14", 4, 3, seed=21)


@pytest.fixture(scope="module")
def vitg_2blocks():
    return _Model("dinov2_vitg14", 2, 1, seed=23)


def _run_extractor(case, mode, hw, n_img, facet, monkeypatch):
    import utilities
    monkeypatch.setenv("ANYLOC_GEMM", mode)
    weights.register_state_dict(case.name, case.sd)
    try:
        ext = utilities.DinoV2ExtractFeatures(case.name, case.layer, facet, device=DEV)
        assert ext.dino_model.gemm == mode
        imgs, ref = case.ref_tokens(hw, n_img, facet)
        x = imgs.to(DEV)
        batched = ext(x).cpu()
        single = torch.cat([ext(x[i:i + 1]) for i in range(n_img)]).cpu()      # the scripts' one-image-per-call pattern
        torch.cuda.synchronize()
    finally:
        weights.unregister_state_dict(case.name)
    T = (hw[0] // 14) * (hw[1] // 14)
    assert batched.shape == single.shape == ref.shape == (n_img, T, ref.shape[-1])
    e_b = float((batched - ref).abs().max())
    e_s = float((single - ref).abs().max())
    print(f"[{case.name} {mode} {hw[0]}x{hw[1]} B={n_img}] token err batched {e_b:.2e}, one image per call {e_s:.2e}")
    assert e_b <= TOKEN_ATOL and e_s <= TOKEN_ATOL, (mode, hw, e_b, e_s)
    assert bool(torch.isfinite(batched).all() and torch.isfinite(single).all())
    # unit rows (norm_descs=True)
    assert float((batched.norm(dim=-1) - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("mode", ["h3", "x6", "f32"])
@pytest.mark.parametrize("hw", [SCRIPT_DEFAULT, VPAIR, DEMO_CAP], ids=["476x630", "588x798", "1022x1022"])
def test_vits_extractor_at_script_shapes(vits_4blocks, hw, mode, monkeypatch):
    """ViT-S/14, 4 blocks, layer-3 'value': the script default, the VPAir shape and the demo cap, batches of 3 and one
    image per call, every GEMM arithmetic, against the restated hub model (non-square / beyond-native positional
    interpolation included)."""
    _run_extractor(vits_4blocks, mode, hw, 3, "value", monkeypatch)


@pytest.mark.parametrize("mode", ["h3", "x6", "f32"])
@pytest.mark.parametrize("hw", [SCRIPT_DEFAULT, DEMO_CAP], ids=["476x630", "1022x1022"])
def test_vitg_extractor_at_script_shapes(vitg_2blocks, hw, mode, monkeypatch):
    """ViT-g/14 geometry (D = 1536, 24 heads, SwiGLU), 2 blocks, layer-1 'value' at 476 x 630 (the script default: T = 1531)
    and 1022 x 1022 (T = 5330: the attention_h3 branch beyond 64 key groups inside the real forward), B = 3 and B = 1."""
    _run_extractor(vitg_2blocks, mode, hw, 3, "value", monkeypatch)


@pytest.mark.parametrize("facet", ["key", "query", "token"])
def test_vitg_other_facets_at_the_script_default(vitg_2blocks, facet, monkeypatch):
    """The other facets of the hook (utilities.py:274-281) at 476 x 630 on the default arithmetic."""
    _run_extractor(vitg_2blocks, "h3", SCRIPT_DEFAULT, 2, facet, monkeypatch)
