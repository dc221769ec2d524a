"""Parity of the token-pooling kernel (csrc/pool.hip through anyloc_pool_tokens) and of the hub-style model
forward (CLS global descriptor) against the CPU restatement of the reference scripts' expressions
(oracle/pool_ref.py: scripts/dino_v2_gp.py:130-135, scripts/dino_v2_gem.py:170-188,
scripts/dino_v2_global_vpr.py:115-128)."""
import pytest
import torch

from anyloc_amd import synth, weights
from oracle import dinov2_ref, pool_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
REL = 1e-5          # fp32 sums of <= 1369 terms in a different order + powf vs torch.pow (~2 ulp each)


def tokens(n_img, n_tok, dim, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(n_img, n_tok, dim, generator=g), dim=-1)


def rel_err(got, want):
    return float((got.cpu() - want).norm(dim=-1).div(want.norm(dim=-1).clamp_min(1e-30)).max())


@pytest.mark.parametrize("shape", [(5, 256, 384), (3, 529, 1536), (2, 1369, 2048), (4, 37, 100), (1, 1, 6), (2, 7, 3)])
def test_average_and_max_match_script_expressions(shape):
    from anyloc_amd import pooling
    x = tokens(*shape, seed=1)
    for method in ("average", "max"):
        want = pool_ref.global_pool(x, method)
        got = pooling.global_pool(x.to(DEV), method)
        assert got.is_cuda and got.shape == want.shape
        if method == "max":
            assert torch.equal(got.cpu(), want)                    # selection: bit-exact
        else:
            assert rel_err(got, want) < REL
    assert not pooling.global_pool(x, "average").is_cuda           # CPU in -> CPU out
    with pytest.raises(NotImplementedError):
        pooling.global_pool(x, "median")


@pytest.mark.parametrize("p", [3, 3.0, 2, 4.5, 0.5])
@pytest.mark.parametrize("use_abs", [False, True])
def test_gem_matches_script_expression(p, use_abs):
    from anyloc_amd import pooling
    x = tokens(6, 529, 1536, seed=2)
    if not use_abs and float(p) != int(p):
        x = x.abs() + 1e-3                      # the reference's own pow gives NaN for t<0 with fractional p
    want = pool_ref.gem_descriptors(x, p, use_abs)
    got = pooling.gem_descriptors(x.to(DEV), p, use_abs)
    assert torch.isfinite(got).all()
    assert rel_err(got, want) < REL, rel_err(got, want)
    if not use_abs and float(p) == 3.0:         # odd power keeps signs: both signs must occur and agree
        assert bool((want < 0).any()) and torch.equal(torch.sign(got.cpu()), torch.sign(want))


def test_gem_fractional_power_of_negative_is_nan_like_torch():
    from anyloc_amd import pooling
    x = tokens(1, 16, 8, seed=3)
    want = pool_ref.gem_descriptors(x, 2.5, False)
    got = pooling.gem_descriptors(x.to(DEV), 2.5, False).cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and bool(torch.isnan(want).any())


def test_ragged_images_and_empty_image():
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(4)
    parts = [torch.randn(n, 384, generator=g) for n in (256, 1, 0, 777)]
    got = ops.pool([p.to(DEV) for p in parts], "average").cpu()
    for i, p in enumerate(parts):
        if p.shape[0]:
            assert float((got[i] - p.mean(0)).abs().max()) < 1e-6
        else:
            assert bool(torch.isnan(got[i]).all())                  # torch.mean over an empty axis
    with pytest.raises(IndexError):
        ops.pool([p.to(DEV) for p in parts], "max")
    got = ops.pool([p.to(DEV) for p in parts if p.shape[0]], "max").cpu()
    assert torch.equal(got, torch.stack([p.max(0)[0] for p in parts if p.shape[0]]))


def test_pool_is_deterministic_and_batch_invariant():
    from anyloc_amd import ops
    x = tokens(40, 529, 1536, seed=5).to(DEV)
    a, b = ops.pool(x, "gem", 3.0), ops.pool(x, "gem", 3.0)
    assert torch.equal(a, b)
    assert torch.equal(ops.pool(x[7:9], "gem", 3.0), a[7:9])


@pytest.mark.parametrize("hw", [(224, 224), (126, 154)])
def test_hub_model_forward_cls_descriptor(hw):
    """``model = torch.hub.load(...)``; ``model(img)`` -> final-norm CLS token (dino_v2_global_vpr.py:115-128)."""
    from anyloc_amd import extractor
    name = "dinov2_vits14"
    sd = synth.synthetic_state_dict(name, 0)
    weights.register_state_dict(name, sd)
    try:
        model = extractor.hub_load("facebookresearch/dinov2", name).eval().to(DEV)
        imgs = torch.cat(synth.synthetic_places(3, 1, hw[0], hw[1], seed=11)[:2])
        got = model(imgs.to(DEV))
        assert got.is_cuda and got.shape == (4, 384)
        want = pool_ref.cls_descriptor(dinov2_ref.build(name, sd), imgs)
        assert float((got.cpu() - want).abs().max()) < 2e-5 * float(want.abs().max())
        assert not model(imgs[:1]).is_cuda
        with pytest.raises(RuntimeError):
            extractor.hub_load("facebookresearch/dino", "dino_vits8")
    finally:
        weights.unregister_state_dict()
