"""Host-side logic of the product that needs no GPU: recall arithmetic, shard merge,
VLAD constructor / cache protocol / error behaviour, k-means iteration control, positional
table, weight resolution, import-time seeding."""
import os

import numpy as np
import pytest
import torch

from anyloc_amd import extractor, kmeans as hip_kmeans, retrieval, synth, weights
from oracle import dinov2_ref, faiss_flat, fpk_kmeans, vlad_ref


def test_import_seeds_rngs(capsys):
    import importlib
    import utilities
    importlib.reload(utilities)
    out = capsys.readouterr().out
    assert "Seed set to: 42" in out
    a = np.random.rand(3)
    np.random.seed(42)
    assert np.array_equal(a, np.random.rand(3))
    for name in ("DinoV2ExtractFeatures", "VLAD", "get_top_k_recall", "seed_everything", "reduce_pca",
                 "CustomDataset", "to_np", "od_down_links"):
        assert hasattr(utilities, name)


def test_recalls_match_oracle_rule():
    rng = np.random.RandomState(0)
    idx = rng.randint(0, 50, size=(17, 20))
    gt = np.empty(17, dtype=object)
    for i in range(17):
        gt[i] = rng.randint(0, 50, size=rng.randint(0, 4))
    top_k = [1, 5, 10, 20]
    a = retrieval.recalls_from_indices(top_k, idx, gt)
    b = vlad_ref.recalls_from_indices(top_k, idx, gt)
    assert a == b
    a2 = retrieval.recalls_from_indices(top_k, idx[:8], gt, use_percentage=False, sub_sample_db=2,
                                        sub_sample_qu=2)
    b2 = vlad_ref.recalls_from_indices(top_k, idx[:8], gt, False, 2, 2)
    assert a2 == b2


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_shard_merge_equals_flat_index(metric):
    g = torch.Generator().manual_seed(3)
    db = torch.randn(103, 16, generator=g)
    db[40] = db[7]                       # exact tie across shards -> lower index first
    qu = torch.randn(9, 16, generator=g)
    k = 12
    d_ref, i_ref = faiss_flat.flat_search(qu, db, k, metric)
    bounds = [0, 30, 31, 80, 103]
    ds, is_ = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        d, i = faiss_flat.flat_search(qu, db[a:b], k, metric)
        ds.append(d.numpy())
        is_.append(np.where(i.numpy() >= 0, i.numpy() + a, -1))
    d, i = retrieval.merge_shard_topk(ds, is_, k, metric)
    assert np.array_equal(i, i_ref.numpy())
    # per-shard GEMMs block differently on the CPU: distances agree to rounding
    np.testing.assert_allclose(d, d_ref.numpy(), rtol=1e-5, atol=1e-5)


def test_vlad_ctor_cache_protocol(tmp_path, capsys):
    import utilities
    v = utilities.VLAD(4, cache_dir=None)
    assert "VLAD caching is disabled." in capsys.readouterr().out
    assert (v.num_clusters, v.desc_dim, v.intra_norm, v.norm_descs, v.mode, v.vlad_mode, v.soft_temp) == \
        (4, None, True, True, "cosine", "hard", 1.0)
    assert v.c_centers is None and v.kmeans is None and v.cache_dir is None
    assert not v.can_use_cache_vlad() and not v.can_use_cache_ids(["a"])
    with pytest.raises(AssertionError):
        utilities.VLAD(4, vlad_mode="medium")
    with pytest.raises(AssertionError):
        v.generate(torch.zeros(3, 8))                 # not fitted (reference utilities.py:948-949)
    cdir = tmp_path / "cache"
    v2 = utilities.VLAD(4, cache_dir=str(cdir))
    assert "Created cache directory" in capsys.readouterr().out and cdir.is_dir()
    assert not v2.can_use_cache_vlad()
    with pytest.raises(ValueError, match="No training descriptors given"):
        v2.fit(None)
    c = torch.randn(4, 8)
    torch.save(c, str(cdir / "c_centers.pt"))
    v3 = utilities.VLAD(4, cache_dir=str(cdir))
    assert "Warning: Cache directory already exists" in capsys.readouterr().out
    v3.fit(None)                                       # restore path: no kernel involved
    out = capsys.readouterr().out
    assert "Using cached cluster centers" in out and "Desc dim set to 8" in out
    assert torch.equal(v3.c_centers, c) and v3.desc_dim == 8 and torch.equal(v3.kmeans.centroids, c)
    assert v3.can_use_cache_vlad()
    assert not v3.can_use_cache_ids(["img0"])
    torch.save(torch.zeros(3, 4, 8), str(cdir / "img0_r.pt"))
    assert v3.can_use_cache_ids("img0", only_residuals=True) and not v3.can_use_cache_ids("img0")
    torch.save(torch.zeros(3, dtype=torch.long), str(cdir / "img0_l.pt"))
    assert v3.can_use_cache_ids(["img0"]) and not v3.can_use_cache_ids(["img0", "img1"])
    assert not v3.can_use_cache_ids(None)


def test_kmeans_host_loop_matches_fpk():
    """The iteration control (init draw, update, NaN->0, tolerance) with the device step
    replaced by the oracle's assign rule must reproduce fast-pytorch-kmeans exactly."""
    def step(x, c, mode, want_labels):
        sim = fpk_kmeans.KMeans.cos_sim(x, c) if mode == "cosine" else fpk_kmeans.KMeans.euc_sim(x, c)
        lab = sim.max(dim=-1)[1]
        onehot = (lab[None, :] == torch.arange(c.shape[0])[:, None]).to(x.dtype)
        return onehot @ x, onehot.sum(-1), lab
    for mode in ("cosine", "euclidean"):
        x = synth.clustered_tokens(1, 3000, 32, n_modes=6, seed=5, noise=0.5)[0]
        np.random.seed(42)
        ref = fpk_kmeans.KMeans(9, mode=mode)         # 9 > 6 modes: exercises empty clusters
        lab_ref = ref.fit_predict(x)
        np.random.seed(42)
        km = hip_kmeans.KMeans(9, mode=mode, step_fn=step)
        lab = km.fit_predict(x)
        assert km.n_iter_ == ref.n_iter_
        assert torch.equal(km.centroids, ref.centroids)
        assert torch.equal(lab, lab_ref)
        assert torch.equal(km.predict(x[:50]), ref.predict(x[:50]))


def test_pos_table_matches_oracle():
    sd = synth.synthetic_state_dict("dinov2_vits14", 0, depth=1)
    for h, w in ((224, 224), (322, 322), (224, 308), (518, 518), (518, 490)):
        a = extractor.interpolate_pos_embed(sd["pos_embed"], h, w)
        b = dinov2_ref.interpolate_pos_embed(sd["pos_embed"], h, w)[0]
        assert a.shape == (1 + (h // 14) * (w // 14), 384)
        assert torch.equal(a, b)


def test_weight_resolution(tmp_path, monkeypatch):
    weights.unregister_state_dict()
    monkeypatch.delenv("ANYLOC_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.setenv("ANYLOC_DINOV2_WEIGHTS", str(tmp_path))
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path / "hub"))
    monkeypatch.setattr(torch.hub, "load_state_dict_from_url",
                        lambda *a, **k: (_ for _ in ()).throw(OSError("offline")))
    with pytest.raises(FileNotFoundError):
        weights.resolve_state_dict("dinov2_vits14")
    with pytest.raises(ValueError):
        weights.resolve_state_dict("dinov2_vitx14")
    sd = synth.synthetic_state_dict("dinov2_vits14", 3, depth=1)
    torch.save(sd, str(tmp_path / "dinov2_vits14_pretrain.pth"))
    got = weights.resolve_state_dict("dinov2_vits14")
    assert torch.equal(got["cls_token"], sd["cls_token"])
    weights.register_state_dict("dinov2_vits14", {"x": 1})
    assert weights.resolve_state_dict("dinov2_vits14") == {"x": 1}
    weights.unregister_state_dict()


def test_synthetic_state_dict_loads_into_hub_layout():
    for name in ("dinov2_vits14", "dinov2_vitg14"):
        sd = synth.synthetic_state_dict(name, 0, depth=2)
        m = dinov2_ref.DinoVisionTransformer(name)
        missing = [k for k in sd if k not in m.state_dict()]
        assert not missing
        for k, v in sd.items():
            assert tuple(m.state_dict()[k].shape) == tuple(v.shape), k


def test_pooling_host_surface(monkeypatch):
    """Host side of the non-VLAD aggregations (argument conventions, error types, device round trip) with the
    device entry point swapped for the CPU restatement; the kernel itself is checked in test_gpu_pooling.py."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _oracle_backend
    from anyloc_amd import pooling
    from oracle import pool_ref
    _oracle_backend.install(monkeypatch)
    x = torch.randn(3, 40, 16, generator=torch.Generator().manual_seed(0))
    assert torch.equal(pooling.global_pool(x, "average"), torch.mean(x, dim=1))
    assert torch.equal(pooling.global_pool(x, "max"), torch.max(x, dim=1)[0])
    with pytest.raises(NotImplementedError, match="ID: median"):
        pooling.global_pool(x, "median")
    g = pooling.gem_descriptors(x, 3)
    m = (x ** 3).mean(1)
    assert torch.allclose(g, m.abs() ** (1 / 3) * m.sign(), atol=1e-6)
    assert torch.equal(pooling.gem_descriptors(x, 3, gem_elem_by_elem=True), g)
    assert torch.allclose(pooling.gem_descriptors(x, 2, gem_use_abs=True), (x.abs() ** 2).mean(1) ** 0.5, atol=1e-6)
    ragged = pooling.global_pool([x[0, :5], x[1]], "average")
    assert ragged.shape == (2, 16) and torch.allclose(ragged[0], x[0, :5].mean(0), atol=1e-6)
    assert torch.equal(pool_ref.gem_descriptors(x, 3, False, True), pool_ref.gem_descriptors(x, 3))


def test_hub_stand_in_refuses_other_repos(monkeypatch):
    with pytest.raises(RuntimeError, match="no network"):
        extractor.hub_load("pytorch/vision", "resnet50")


def _decaying(n, f, seed, rank=None, decay=0.8):
    """Rows with a well-separated, geometrically decaying spectrum (unique principal axes)."""
    g = torch.Generator().manual_seed(seed)
    r = min(n, f) if rank is None else rank
    q1, _ = torch.linalg.qr(torch.randn(n, r, generator=g, dtype=torch.float64))
    q2, _ = torch.linalg.qr(torch.randn(f, r, generator=g, dtype=torch.float64))
    s = 10.0 * decay ** torch.arange(r, dtype=torch.float64)
    return ((q1 * s) @ q2.t() + 0.3 * torch.randn(1, f, generator=g, dtype=torch.float64)).float()


@pytest.mark.parametrize("shape,k,whiten", [((60, 200), 16, False), ((60, 200), 16, True), ((300, 50), 12, False),
                                            ((300, 50), 12, True), ((40, 41), 7, False)])
def test_pca_host_logic_matches_sklearn(monkeypatch, shape, k, whiten):
    """anyloc_amd.pca against the reference's own implementation (sklearn PCA, svd_solver='full') with the GEMM
    entry point swapped for torch on the CPU: Gram / scatter branch choice, eigen-ordering, svd_flip signs,
    whitening scale, transform order."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _oracle_backend
    from sklearn.decomposition import PCA as SkPCA
    from anyloc_amd import pca
    _oracle_backend.install(monkeypatch)
    x, y = _decaying(*shape, seed=1), _decaying(17, shape[1], seed=2)
    sk = SkPCA(k, svd_solver="full", whiten=whiten)
    want_tr, want_ts = sk.fit_transform(x.double().numpy()), None
    want_ts = sk.transform(y.double().numpy())
    ours = pca.PCA(k, whiten=whiten)
    got_tr, got_ts = ours.fit_transform(x), ours.transform(y)
    assert np.abs(ours.components_.numpy() - sk.components_).max() < 2e-5
    assert np.allclose(ours.explained_variance_.numpy(), sk.explained_variance_, rtol=1e-5)
    assert np.allclose(ours.singular_values_.numpy(), sk.singular_values_, rtol=1e-5)
    assert np.allclose(ours.mean_.numpy(), sk.mean_, atol=1e-6)
    scale = np.abs(want_tr).max()
    assert np.abs(got_tr.numpy() - want_tr).max() < 1e-4 * scale
    assert np.abs(got_ts.numpy() - want_ts).max() < 1e-4 * max(scale, np.abs(want_ts).max())


def test_reduce_pca_surface_matches_reference_function(monkeypatch, capsys):
    """utilities.reduce_pca (numpy in / numpy out, low_factor and fallback branches) against the reference's own
    function imported verbatim (utilities.py:522-586)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _oracle_backend
    from oracle import ref_loader
    if not ref_loader.reference_available():
        pytest.skip("reference tree not present")
    import utilities
    ref = ref_loader.load_reference_utilities()
    _oracle_backend.install(monkeypatch)
    tr, ts = _decaying(80, 48, seed=3, decay=0.92).numpy(), _decaying(9, 48, seed=4, decay=0.92).numpy()
    for kwargs in (dict(lower_dim=10), dict(lower_dim=10, whitening=True), dict(lower_dim=10, low_factor=0.3)):
        a_tr, a_ts = utilities.reduce_pca(tr, ts, **kwargs)
        b_tr, b_ts = ref.reduce_pca(tr.copy(), ts.copy(), **kwargs)
        assert isinstance(a_tr, np.ndarray) and a_tr.shape == b_tr.shape and a_ts.shape == b_ts.shape
        m = np.abs(b_tr).max()
        assert np.abs(a_tr - b_tr).max() < 2e-4 * m and np.abs(a_ts - b_ts).max() < 2e-4 * m, kwargs
    # too few samples: joint projection to `fallback` dims first (reference :566-574)
    tr, ts = _decaying(30, 64, seed=5, rank=20).numpy(), _decaying(8, 64, seed=6, rank=8).numpy()
    a_tr, a_ts = utilities.reduce_pca(tr, ts, lower_dim=6, low_factor=0.5, fallback=16)
    b_tr, b_ts = ref.reduce_pca(tr.copy(), ts.copy(), lower_dim=6, low_factor=0.5, fallback=16)
    assert "Too few samples, fallback to 16d first" in capsys.readouterr().out
    assert a_tr.shape == b_tr.shape == (30, 6) and a_ts.shape == (8, 6)
    # the lowest-variance axes of a rank-deficient matrix are not unique: compare the well-defined top half
    m = np.abs(b_tr[:, :3]).max()
    assert np.abs(a_tr[:, :3] - b_tr[:, :3]).max() < 5e-4 * m


def test_lead_plans_of_the_batched_layernorm_role_pass_the_host_check():
    """csrc/tile_order.hpp LeadPlan (ABI 9 diagnostic, host only): for every shape the batched forward can meet -- tile rows x tile
    columns x scheduling group x row count -- the simulated launch holds every GEMM tile once, normalises every row once and puts
    every producer ahead of its consumers in workgroup-id order; shapes with fewer than 8 scheduling groups are refused (the
    forward keeps two launches)."""
    import ctypes as C
    from anyloc_amd import _lib
    lib = _lib.load()
    g = C.c_uint32()
    for M in (32330, 8480, 8192, 8193, 109600, 16165, 5300, 64 * 128 - 1, 12345, 1370 * 80, 257 * 40, 1531 * 8):
        tm = (M + 127) // 128
        for tn in (12, 16, 18, 32, 6, 7, 33):
            for gm in (8, 4, 1, 16, 3):
                ok = lib.anyloc_h3_lead_plan_check(tm, tn, gm, M, C.byref(g))
                assert bool(ok) == (tm >= 8 * gm), (M, tn, gm, ok)
                if ok:
                    leads = (M + 15) // 16
                    assert tm * tn + leads <= g.value < tm * tn + leads + 8 * 16 * gm + 64, (M, tn, gm, g.value)
    assert lib.anyloc_h3_lead_plan_check(70, 18, 8, 70 * 128 + 5, C.byref(g)) == 0          # rows and tile rows disagree
    assert lib.anyloc_h3_lead_plan_check(0, 18, 8, 100, None) == 0
