"""Parity of the HIP DINOv2 forward (csrc/vit.hip through the C ABI) and of the whole
config-1 pipeline against the golden vectors recorded from the reference's own code."""
import os

import numpy as np
import pytest
import torch

from anyloc_amd import synth, weights
from oracle import dinov2_ref, vlad_ref
from oracle.make_golden import probe_vector

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOKEN_ATOL = 2e-5        # unit-norm token rows, fp32 end to end through <= 12 blocks


@pytest.fixture(scope="module")
def g1(golden_dir):
    return np.load(os.path.join(golden_dir, "config1_vits14_l9_value_k8.npz"))


@pytest.fixture(scope="module")
def c1(g1):
    sd = synth.synthetic_state_dict(str(g1["model"]), int(g1["weights_seed"]))
    weights.register_state_dict(str(g1["model"]), sd)
    db, qu, gt = synth.synthetic_places(int(g1["n_db"]), int(g1["n_qu"]), int(g1["hw"]), int(g1["hw"]),
                                        seed=int(g1["images_seed"]))
    yield sd, torch.cat([db, qu]), gt
    weights.unregister_state_dict()


def test_tokens_match_reference_golden(g1, c1):
    import utilities
    sd, imgs, _ = c1
    ext = utilities.DinoV2ExtractFeatures("dinov2_vits14", 9, "value", device=DEV)
    one = ext(imgs[:1].to(DEV))                          # the reference's B=1 calling convention
    assert one.shape == (1, 256, 384) and one.is_cuda
    assert float((one[0].cpu() - torch.from_numpy(g1["tokens_img0"])).abs().max()) < TOKEN_ATOL
    allt = ext(imgs.to(DEV)).cpu()                       # batched: 32 images in one launch sequence
    assert float((allt[31] - torch.from_numpy(g1["tokens_img31"])).abs().max()) < TOKEN_ATOL
    assert float((allt[0] - one[0].cpu()).abs().max()) < 1e-6      # batch-size invariance
    pv = probe_vector(384)
    assert float(((allt @ pv) - torch.from_numpy(g1["token_proj"])).abs().max()) < 2e-4
    np.testing.assert_allclose((allt.double() ** 2).sum(-1).numpy(), g1["token_sumsq"], atol=1e-5)
    cpu_in = ext(imgs[:2])                               # CPU tensor in -> CPU tensor out
    assert cpu_in.device.type == "cpu" and float((cpu_in - allt[:2]).abs().max()) < 1e-6


def test_facet_variants_match_golden(g1, c1):
    import utilities
    _, imgs, _ = c1
    pv = probe_vector(384)
    cases = {"query": dict(layer=9, facet="query"), "key": dict(layer=9, facet="key"),
             "token": dict(layer=9, facet="token"),
             "value_cls_raw": dict(layer=9, facet="value", use_cls=True, norm_descs=False),
             "token_l11": dict(layer=11, facet="token")}
    for name, kw in cases.items():
        ext = utilities.DinoV2ExtractFeatures("dinov2_vits14", device=DEV, **kw)
        out = ext(imgs[:1].to(DEV))[0].cpu()
        assert tuple(out.shape) == tuple(g1[f"facet_{name}_shape"]), name
        scale = max(1.0, float(np.abs(g1[f"facet_{name}_proj"]).max()))
        assert float(((out @ pv) - torch.from_numpy(g1[f"facet_{name}_proj"])).abs().max()) < 3e-4 * scale, name
        ref_head = torch.from_numpy(g1[f"facet_{name}_head"])
        assert float((out[:4, :16] - ref_head).abs().max()) < 3e-5 * max(1.0, float(ref_head.abs().max())), name


def test_non_square_and_multitap_vs_oracle(c1):
    import utilities
    sd, imgs, _ = c1
    model = dinov2_ref.build("dinov2_vits14", sd)
    g = torch.Generator().manual_seed(77)
    img = torch.randn(2, 3, 224, 308, generator=g)       # 16 x 22 patches: pos-embed interpolation
    ext = utilities.DinoV2ExtractFeatures("dinov2_vits14", 9, "value", device=DEV)
    out = ext(img.to(DEV)).cpu()
    ref = dinov2_ref.extract_facet(model, img, 9, "value")
    assert out.shape == ref.shape == (2, 352, 384)
    assert float((out - ref).abs().max()) < TOKEN_ATOL
    # two taps in one forward == two reference extractors, concat on the feature axis, renormalise
    multi = ext.extract_multi(img.to(DEV), [5, 9], "value").cpu()
    r5 = dinov2_ref.extract_facet(model, img, 5, "value")
    cat = torch.nn.functional.normalize(torch.cat([r5, ref], dim=-1), dim=-1)
    assert multi.shape == (2, 352, 768)
    assert float((multi - cat).abs().max()) < TOKEN_ATOL
    with pytest.raises(AssertionError):
        ext(torch.zeros(1, 3, 225, 224, device=DEV))


def test_vitg_swiglu_blocks_vs_oracle():
    """ViT-g/14 geometry (D=1536, 24 heads, SwiGLU 4096) at 322x322, truncated to 3 blocks so
    the CPU oracle finishes in seconds; hook layer 2 'value' (partial-QKV early exit) and
    layer 1 'token' (full block incl. SwiGLU)."""
    import utilities
    sd = synth.synthetic_state_dict("dinov2_vitg14", 1, depth=3)
    weights.register_state_dict("dinov2_vitg14", sd)
    try:
        full = dinov2_ref.DinoVisionTransformer("dinov2_vitg14")
        full.blocks = full.blocks[:3]
        full.load_state_dict(sd, strict=True)
        full.eval()
        g = torch.Generator().manual_seed(5)
        img = torch.randn(2, 3, 322, 322, generator=g)
        for layer, facet in ((2, "value"), (1, "token")):
            ext = utilities.DinoV2ExtractFeatures("dinov2_vitg14", layer, facet, device=DEV)
            out = ext(img.to(DEV)).cpu()
            ref = dinov2_ref.extract_facet(full, img, layer, facet)
            assert out.shape == (2, 529, 1536)
            assert float((out - ref).abs().max()) < TOKEN_ATOL, (layer, facet)
    finally:
        weights.unregister_state_dict("dinov2_vitg14")


def test_config1_pipeline_through_reference_surface(g1, c1, capsys):
    """BASELINE.json configs[0] end to end on the HIP path, driven exactly like
    scripts/dino_v2_vlad.py drives the reference (B=1 extraction, .cpu(), VLAD.fit on the db
    tokens, generate_multi, get_top_k_recall) -- compared with the reference's recorded outputs."""
    import utilities
    _, imgs, gt = c1
    n_db, K = int(g1["n_db"]), int(g1["K"])
    utilities.seed_everything(42)
    vlad = utilities.VLAD(K, None, cache_dir=None)
    ext = utilities.DinoV2ExtractFeatures("dinov2_vits14", 9, "value", device=DEV)
    toks = torch.cat([ext(im[None].to(DEV)).cpu() for im in imgs])
    vlad.fit(toks[:n_db].reshape(-1, toks.shape[-1]))
    centers_ref = torch.from_numpy(g1["centers"])
    # k-means on 6144 tokens is chaotic in the last bits; compare what matters downstream
    rel_c = float((vlad.c_centers - centers_ref).norm() / centers_ref.norm())
    print("kmeans iters", vlad.kmeans.n_iter_, "golden", int(g1["kmeans_iters"]), "centre rel err", rel_c)
    same_vocab = rel_c < 1e-4
    if not same_vocab:
        # vocabulary diverged by a token flipping cluster mid-fit: pin the vocabulary instead
        vlad.c_centers = centers_ref
        vlad.kmeans.centroids = centers_ref
    labels = torch.stack([vlad.kmeans.predict(t) for t in toks])
    lab_ref = torch.from_numpy(g1["labels"].astype(np.int64))
    flips = (labels != lab_ref)
    if flips.any():
        sc = vlad_ref.fpk_cosine_scores(toks[flips], centers_ref)
        top2 = sc.topk(2, dim=1)[0]
        assert float((top2[:, 0] - top2[:, 1]).max()) < 2e-5
    db_vlads = vlad.generate_multi(toks[:n_db])
    qu_vlads = vlad.generate_multi(toks[n_db:])
    vl = torch.cat([db_vlads, qu_vlads])
    ref = torch.from_numpy(g1["vlads"])
    rel = ((vl - ref).norm(dim=1) / ref.norm(dim=1))
    print("VLAD rel err max", float(rel.max()), "label flips", int(flips.sum()))
    assert float(rel.max()) < (1e-4 if flips.any() else 2e-5)
    top_k = list(range(1, 21))
    d, i, r = utilities.get_top_k_recall(top_k, db_vlads, qu_vlads, gt)
    assert [r[k] for k in top_k] == list(g1["recalls"])                 # Recall@k identical
    assert np.array_equal(i.numpy()[:, :5], g1["top_idx"][:, :5])
    np.testing.assert_allclose(d.numpy(), g1["top_dist"], atol=2e-5)
    capsys.readouterr()


@pytest.mark.parametrize("batch", [1, 2])
def test_small_batch_kernels_are_bitwise_the_plain_ones(batch, monkeypatch):
    """Round-3 small-batch kernels (option h3s_enable = 0): one or two images run 64x64 GEMM tiles with four / two k-blocks
    per ring stage (proj, fc2) and a LayerNorm with one row per wave: scheduling changes only -- the tokens must equal, bit
    for bit, those of the kernels with one k-block per stage and four rows per wave (options h3_deep_max = h3_deep2_max =
    ln_small_rows = 0)."""
    import utilities
    name = "dinov2_vitg14"
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 3, device=DEV, depth=3))
    try:
        ext = utilities.DinoV2ExtractFeatures(name, 2, "value", device=DEV)
        img = torch.randn(batch, 3, 322, 322, generator=torch.Generator().manual_seed(batch)).to(DEV)
        from anyloc_amd import ops
        with ops.options(h3s_enable=0):
            got = ext(img).clone()
            with ops.options(h3_deep_max=0, h3_deep2_max=0, ln_small_rows=0):
                want = ext(img).clone()
        assert torch.isfinite(got).all() and torch.equal(got, want)
    finally:
        weights.unregister_state_dict(name)


@pytest.mark.parametrize("batch", [1, 2, 3, 5])
def test_small_m_plans_agree_with_the_plain_kernels(batch):
    """The small-M plans of csrc/gemm_h3s.hip (other tile shapes, several k-blocks per ring stage, split-K with a
    deterministic split-order reduction) change the summation order over k, nothing else: every plan -- the table's choice
    and each forced (tile configuration, ring depth, split factor) -- gives tokens within 2e-6 of the round-3 kernels, the
    same bits run to run, on a 3-block ViT-g (all four block GEMMs + the facet GEMM go through the plans)."""
    import utilities
    from anyloc_amd import ops
    name = "dinov2_vitg14"
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 3, device=DEV, depth=3))
    try:
        ext = utilities.DinoV2ExtractFeatures(name, 2, "value", device=DEV)
        ext.dino_model.ffn_check = False
        img = torch.randn(batch, 3, 322, 322, generator=torch.Generator().manual_seed(10 + batch)).to(DEV)
        with ops.options(h3s_enable=0):
            want = ext(img).clone()
        got = ext(img).clone()
        assert torch.isfinite(got).all()
        assert float((got - want).abs().max()) <= 2e-6
        assert torch.equal(got, ext(img))
        with ops.options(ln_direct_rows=0):                              # the single-wave LayerNorm of few-row calls: same bits as the tiled one
            assert torch.equal(got, ext(img))
        plans = [(c, kb, ks, st) for c in range(8) for kb, ks, st in ((1, 1, 3), (2, 3, 6), (4, 2, 3), (1, 8, 6), (2, 5, 3))]
        for cfg, kb, ks, st in plans:
            with ops.options(h3s_cfg=cfg, h3s_kb=kb, h3s_ksplit=ks, h3s_stages=st):
                a = ext(img).clone()
                b = ext(img)
            assert float((a - want).abs().max()) <= 2e-6, (cfg, kb, ks, st, float((a - want).abs().max()))
            assert torch.equal(a, b), (cfg, kb, ks, st, "not reproducible")
    finally:
        weights.unregister_state_dict(name)


def test_split_k_hand_off_under_load():
    """The split-K hand-off of gemm_h3_kernel.hpp (write-through sc1 slab stores, every wave drained, ONE relaxed agent-scope
    ticket, the last arrival reads the slabs back with sc1 loads in split order) under load: a 3-block ViT-g forward at one
    and at two images with split-K forced on EVERY block GEMM (64 x 64 tiles: up to 1 152 concurrent tiles x 8 splits per
    launch, 15 launches per forward), 25 forwards per plan.  A stale slab read -- the failure this recipe could have --
    shows as a run that differs from the first one: every run must be bit-identical, and within the k-order bar of the
    unsplit plan."""
    import utilities
    from anyloc_amd import ops
    name = "dinov2_vitg14"
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 4, device=DEV, depth=3))
    try:
        ext = utilities.DinoV2ExtractFeatures(name, 2, "value", device=DEV)
        ext.dino_model.ffn_check = False
        for batch in (1, 2):
            img = torch.randn(batch, 3, 322, 322, generator=torch.Generator().manual_seed(70 + batch)).to(DEV)
            with ops.options(h3s_cfg=0, h3s_kb=1, h3s_ksplit=1, h3s_stages=3):
                unsplit = ext(img).clone()
            for ks, st in ((8, 3), (3, 6), (2, 6)):
                with ops.options(h3s_cfg=0, h3s_kb=1, h3s_ksplit=ks, h3s_stages=st):
                    first = ext(img).clone()
                    assert float((first - unsplit).abs().max()) <= 2e-6, (batch, ks)
                    for rep in range(25):
                        assert torch.equal(ext(img), first), (batch, ks, st, rep, "a split-K run differs from the first one")
    finally:
        weights.unregister_state_dict(name)


def test_ffn_bound_telemetry_switches_a_loose_block_to_the_exact_quantiser():
    """h3 forward: the fused fc1 epilogue quantises the hidden activation against a Cauchy-Schwarz bound.  A weight set
    whose bound is far above the real activations (one fc1 row of huge norm along the direction LayerNorm's output never
    moves in) trips the telemetry: the block is switched to the exact row-maximum quantiser and the tokens still meet the
    oracle bar; ordinary blocks stay fused."""
    import utilities
    from anyloc_amd import extractor as ex
    from oracle import dinov2_ref
    name = "dinov2_vits14"
    sd = synth.synthetic_state_dict(name, 5, device="cpu", depth=4)
    # block 2, hidden unit 7: weight row c * d with d . (LN2 output - LN2 bias) = 0 for every token (d ~ 1 / norm2.weight, the
    # one direction a LayerNorm output cannot move along) and bias -c * (d . LN2 bias): a huge row norm -- the bound of every
    # token row grows ~3000 x -- whose pre-activation is ~0 for every token, so the real activations do not change
    w, b = sd["blocks.2.norm2.weight"].double(), sd["blocks.2.norm2.bias"].double()
    d = (1.0 / w) / (1.0 / w).norm()
    f1 = sd["blocks.2.mlp.fc1.weight"].double()
    c = 3000.0 * float(f1.norm(dim=1).max())
    f1[7] = c * d
    sd["blocks.2.mlp.fc1.weight"] = f1.float()
    sd["blocks.2.mlp.fc1.bias"][7] = float(-c * (d * b).sum())
    weights.register_state_dict(name, {k: v.to(DEV) for k, v in sd.items()})
    full = dinov2_ref.DinoVisionTransformer(name)
    full.blocks = full.blocks[:4]
    full.load_state_dict(sd, strict=True)
    img = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    ref = dinov2_ref.extract_facet(full.eval(), img, 3, "token")

    def run():
        ext = utilities.DinoV2ExtractFeatures(name, 3, "token", device=DEV)
        assert ext.dino_model.gemm == "h3"
        got = ext(img.to(DEV)).cpu()
        m = ext.dino_model
        assert m.ffn_looseness is not None and m.ffn_looseness[2] > ex.FFN_LOOSENESS_MAX, m.ffn_looseness
        assert m.ffn_exact_blocks == {2}, (m.ffn_exact_blocks, m.ffn_looseness)
        assert all(0 < m.ffn_looseness[i] <= ex.FFN_LOOSENESS_MAX for i in (0, 1, 3)), m.ffn_looseness
        assert m.ffn_reruns == 3                                  # every image of the call trips block 2 and was run again
        assert float((got - ref).abs().max()) <= 2e-5
        # nothing is sticky (round 6): the next call decides again from its own data and gives the same bits
        again = ext(img.to(DEV)).cpu()
        assert torch.equal(again, got)
        assert m.ffn_reruns == 6 and m.ffn_exact_blocks == {2}
        return got, m.ffn_looseness.copy()
    try:
        a = run()
        b2 = run()                                                   # a second handle: the same figures and bits
        assert torch.equal(a[0], b2[0]) and np.array_equal(a[1], b2[1])
    finally:
        weights.unregister_state_dict(name)


def test_ffn_bound_decision_is_per_image_and_leaves_no_state(monkeypatch):
    """Round 6: the decision between the bound quantiser and the exact one is taken per CALL and per IMAGE from the
    looseness the forward itself measured (one figure per block and image).  With the threshold moved between the images'
    own figures, some images of a batch trip a block and others do not: every image's tokens are bit for bit what the image
    gives at the same position of a call whose other images are clean -- whatever its batch mates contain -- and the same
    before and after a tripping image went through the handle."""
    import utilities
    from anyloc_amd import extractor as ex
    from oracle import dinov2_ref
    name = "dinov2_vits14"
    sd = synth.synthetic_state_dict(name, 9, device="cpu", depth=4)
    weights.register_state_dict(name, {k: v.to(DEV) for k, v in sd.items()})
    try:
        ext = utilities.DinoV2ExtractFeatures(name, 3, "token", device=DEV)
        m = ext.dino_model
        g = torch.Generator().manual_seed(21)
        imgs = torch.cat([torch.randn(3, 3, 224, 224, generator=g), 0.05 * torch.randn(3, 3, 224, 224, generator=g)]).to(DEV)
        # per-image figures of block 1 at the stock threshold (nothing trips on ordinary weights)
        ext(imgs)
        assert m.ffn_reruns == 0 and m.ffn_exact_blocks == set()
        per_img = m._telemetry[:4 * 6].cpu().reshape(4, 6)
        assert float(per_img.min()) > 1.0 and float(per_img.max()) <= ex.FFN_LOOSENESS_MAX
        fig = per_img.max(dim=0).values                                            # an image trips when ANY of its blocks does
        order = torch.argsort(fig)
        thr = float(0.5 * (fig[order[2]] + fig[order[3]]))                         # three images below, three above
        if not fig[order[2]] < thr < fig[order[3]]:
            pytest.skip("the images' looseness figures coincide: no threshold separates them")
        monkeypatch.setattr(ex, "FFN_LOOSENESS_MAX", thr)
        trips = [bool((m_l > thr).any()) for m_l in per_img.t()]
        assert any(trips) and not all(trips), trips
        clean = trips.index(False)
        # what image i gives at batch position i of a six-image call whose other images are all the clean one (the kernels'
        # summation orders depend on the call's row count and on the position, so both are kept)
        def filler_with(i):
            b = imgs[clean:clean + 1].repeat(6, 1, 1, 1)
            b[i] = imgs[i]
            return b
        want = [ext(filler_with(i))[i].clone() for i in range(6)]
        runs0 = m.ffn_reruns
        batch = ext(imgs)
        assert m.ffn_reruns - runs0 == sum(trips) and m.ffn_exact_blocks, (m.ffn_reruns, runs0, trips)
        for i in range(6):
            assert torch.equal(batch[i], want[i]), (i, trips)
        # a clean call after a tripping one through the same handle: the same bits as before it, and nothing stays switched
        again = ext(filler_with(clean))
        assert torch.equal(again[clean], want[clean]) and m.ffn_exact_blocks == set()
        # both quantisers meet the oracle bar
        full = dinov2_ref.DinoVisionTransformer(name)
        full.blocks = full.blocks[:4]
        full.load_state_dict(sd, strict=True)
        ref = dinov2_ref.extract_facet(full.eval(), imgs.cpu(), 3, "token")
        assert float((batch.cpu() - ref).abs().max()) <= 2e-5
    finally:
        weights.unregister_state_dict(name)
