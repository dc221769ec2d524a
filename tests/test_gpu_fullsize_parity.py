"""Full-size oracle parity of the product path, in every block-GEMM arithmetic (GPU box only).

BASELINE.json configs[1] geometry at FULL depth -- DINOv2 ViT-g/14, 32 blocks, 322x322, layer-31 'value',
K=32 VLAD, top-k -- and configs[4] (ViT-L/14 518x518, taps 20+23) against the CPU oracle
(oracle/dinov2_ref.py = restated reference forward, reference utilities.py:263-285), for the three GEMM
modes the extractor can run in ("h3" two-term fp16 splits = default, "x6" three-way bf16 splits, "f32"
fp32 MFMA), on ordinary synthetic weights AND on the outlier-weight stress of
``synth.outlier_state_dict`` (heavy-tailed LayerNorm gains, a massive residual channel, LayerScale gammas
over five decades, a register-like token).

Tolerances are the north_star's: unit-norm tokens <= 2e-5 max-abs vs the fp32 oracle, VLAD <= 1e-5
L2-relative, cluster ids identical except where the oracle's own top-2 cosine gap is < 1e-6 (an fp32
tie), top-k indices identical.  In addition every mode must be as close to a float64 evaluation of the
same network as the fp32 CPU oracle itself is (factor 3 + 1e-7): the split arithmetics claim fp32-level
accuracy, this is where the claim is checked at full depth.
"""
import numpy as np
import pytest
import torch

from anyloc_amd import synth, weights
from oracle import dinov2_ref, faiss_flat, vlad_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
MODES = ("h3", "x6", "f32")
TOKEN_ATOL = 2e-5
VLAD_RTOL = 1e-5
GAP_TOL = 1e-6
K = 32


def _oracle_tokens(name, sd, depth, imgs, taps, dtype):
    model = dinov2_ref.DinoVisionTransformer(name)
    model.blocks = model.blocks[:depth]
    model.load_state_dict(sd, strict=True)
    model.eval()
    if dtype == torch.float64:
        model = model.double()
    outs = []
    for im in imgs:                                       # B=1, the reference's calling convention
        per_tap = [dinov2_ref.extract_facet(model, im[None].to(dtype), l, f)[0] for l, f in taps]
        t = torch.cat(per_tap, dim=-1)
        outs.append(torch.nn.functional.normalize(t, dim=-1) if len(taps) > 1 else t)
    return torch.stack(outs)


class Case:
    """Oracle side of one (architecture, weights) case, computed once per module."""

    def __init__(self, name, depth, hw, taps, n_img, seed, stress, n_clusters=K):
        torch.set_num_threads(max(1, min(32, torch.get_num_threads() * 2)))
        self.name, self.depth, self.taps = name, depth, taps
        sd = synth.synthetic_state_dict(name, seed, depth=depth)
        self.sd = synth.outlier_state_dict(sd, name, seed + 1) if stress else sd
        db, qu, _ = synth.synthetic_places(n_img // 2, n_img - n_img // 2, hw, hw, seed=seed + 2)
        self.imgs = torch.cat([db, qu])
        self.tok32 = _oracle_tokens(name, self.sd, depth, self.imgs, taps, torch.float32)
        self.tok64 = _oracle_tokens(name, self.sd, depth, self.imgs, taps, torch.float64)
        self.err32 = float((self.tok32.double() - self.tok64).abs().max())
        D = self.tok32.shape[-1]
        g = torch.Generator().manual_seed(seed + 3)
        flat = self.tok32.reshape(-1, D)
        self.centers = (0.8 * flat[torch.randperm(flat.shape[0], generator=g)[:n_clusters]]).contiguous()
        both = [vlad_ref.vlad_hard(t, self.centers) for t in self.tok32]
        self.vlads = torch.stack([b[0] for b in both])
        self.labels = torch.stack([b[1] for b in both])
        # database: random unit-block rows + the oracle VLADs of the first half of the images (the "places")
        rnd = torch.nn.functional.normalize(torch.randn(600, n_clusters, D, generator=g), dim=-1) / n_clusters ** 0.5
        self.db = torch.cat([self.vlads[:n_img // 2], rnd.reshape(600, -1)])
        qn, dbn = torch.nn.functional.normalize(self.vlads), torch.nn.functional.normalize(self.db)
        self.top_d, self.top_i = faiss_flat.flat_search(qn, dbn, 10)
        # the same scores summed in float64 (a 131 072-term fp32 dot product of the oracle is itself ~3e-6 off)
        self.top_d64 = torch.gather(qn.double() @ dbn.double().t(), 1, self.top_i)


def _run_mode(case, mode, monkeypatch):
    import utilities
    from anyloc_amd import ops, retrieval
    monkeypatch.setenv("ANYLOC_GEMM", mode)
    weights.register_state_dict(case.name, case.sd)
    ext = None
    try:
        ext = utilities.DinoV2ExtractFeatures(case.name, case.taps[-1][0], case.taps[-1][1], device=DEV)
        assert ext.dino_model.gemm == mode
        x = case.imgs.to(DEV)
        if len(case.taps) == 1:
            tok = ext(x)
        else:
            tok = ext.extract_multi(x, [l for l, _ in case.taps], case.taps[0][1])
        v, lab = ops.vlad(tok, case.centers.to(DEV), return_labels=True)
        d, i = retrieval.search(case.db.to(DEV), v, 10)
        torch.cuda.synchronize()
        return tok.cpu(), v.cpu(), lab.cpu().reshape(len(case.imgs), -1), d.cpu(), i.cpu()
    finally:
        weights.unregister_state_dict(case.name)
        del ext


def _check(case, mode, tok, v, lab, d, i):
    err = float((tok - case.tok32).abs().max())
    err64 = float((tok.double() - case.tok64).abs().max())
    print(f"[{case.name} {mode}] token err vs fp32 oracle {err:.2e}, vs float64 {err64:.2e} "
          f"(fp32 oracle vs float64 {case.err32:.2e})")
    assert tok.shape == case.tok32.shape
    assert err <= TOKEN_ATOL, (mode, err)
    assert err64 <= 3.0 * case.err32 + 1e-7, (mode, err64, case.err32)
    flips = lab != case.labels
    if flips.any():
        sc = vlad_ref.fpk_cosine_scores(case.tok32[flips], case.centers).topk(2, dim=1)[0]
        gap = float((sc[:, 0] - sc[:, 1]).max())
        assert gap < GAP_TOL, f"{mode}: {int(flips.sum())} cluster-id flips, largest oracle gap {gap:.3e}"
    for n in range(len(case.imgs)):
        ref = case.vlads[n]
        if flips[n].any():       # an fp32 tie moved one token: score the descriptor under the same assignment
            ref = vlad_ref.vlad_hard(case.tok32[n], case.centers, labels=lab[n])[0]
        rel = float((v[n].double() - ref.double()).norm() / ref.double().norm())
        assert rel <= VLAD_RTOL, (mode, n, rel)
    clean = ~flips.any(dim=1)
    assert torch.equal(i[clean], case.top_i[clean]), mode
    assert torch.equal(i[:, 0], case.top_i[:, 0]), mode
    np.testing.assert_allclose(d[clean].numpy(), case.top_d[clean].numpy(), atol=1e-5)
    np.testing.assert_allclose(d[clean].double().numpy(), case.top_d64[clean].numpy(), atol=2e-6)


@pytest.fixture(scope="module")
def vitg_plain():
    return Case("dinov2_vitg14", 32, 322, [(31, "value")], 4, seed=3, stress=False)


@pytest.fixture(scope="module")
def vitg_stress():
    return Case("dinov2_vitg14", 32, 322, [(31, "value")], 4, seed=7, stress=True)


@pytest.fixture(scope="module")
def vitl_taps():
    return Case("dinov2_vitl14", 24, 518, [(20, "value"), (23, "value")], 2, seed=11, stress=False, n_clusters=64)


@pytest.mark.parametrize("mode", MODES)
def test_vitg_full_depth_vs_oracle(vitg_plain, mode, monkeypatch):
    """configs[1]: ViT-g/14, 32 blocks, 322x322, L31 'value' -> VLAD K=32 -> top-10, vs the CPU oracle."""
    _check(vitg_plain, mode, *_run_mode(vitg_plain, mode, monkeypatch))


@pytest.mark.parametrize("mode", MODES)
def test_vitg_full_depth_outlier_weights(vitg_stress, mode, monkeypatch):
    """The same with outlier weights (synth.outlier_state_dict): the row-scaled 22-bit split ("h3") must
    hold the fp32 bar when rows span several decades."""
    hot = vitg_stress.sd["blocks.5.norm1.weight"].abs()
    assert float(hot.max() / hot.median()) > 500          # the stress really is in the weights
    _check(vitg_stress, mode, *_run_mode(vitg_stress, mode, monkeypatch))


@pytest.mark.parametrize("mode", MODES)
def test_vitl_518_two_taps_vs_oracle(vitl_taps, mode, monkeypatch):
    """configs[4] geometry: ViT-L/14 (GELU MLP), 518x518 = 1369 patches, taps L20 + L23 'value' in ONE
    forward, concatenated and re-normalised (reference scripts/dino_v2_vlad_viz.py:175-196), K=64 VLAD."""
    assert vitl_taps.tok32.shape == (2, 1369, 2048)
    _check(vitl_taps, mode, *_run_mode(vitl_taps, mode, monkeypatch))


def test_fused_and_unfused_h3_agree(vitg_plain, monkeypatch):
    """A/B switch of the fused fp16-plane producers (option h3_fuse = 0: fp32 activations + separate
    quantiser passes): both meet the oracle bar and agree with each other far inside it."""
    t1 = _run_mode(vitg_plain, "h3", monkeypatch)[0]
    from anyloc_amd import ops
    ops.set_option("h3_fuse", 0)
    t0 = _run_mode(vitg_plain, "h3", monkeypatch)[0]
    assert float((t0 - vitg_plain.tok32).abs().max()) <= TOKEN_ATOL
    assert float((t1 - t0).abs().max()) <= 2e-6
