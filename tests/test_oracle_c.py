"""The C restatement (oracle/c/anyloc_oracle.c, double accumulators) against the golden vectors recorded from the
REFERENCE's own code (tests/golden/*.npz, oracle/make_golden.py) and against the torch restatements.  CPU only.

Bars (the C restatement is exact arithmetic on the fp32 inputs; the recordings are the reference's fp32 results):
labels identical except where the exact top-2 gap is < 1e-6, descriptors <= 2e-6 max-abs on unit vectors, top-k indices
identical, distances <= 2e-6, k-means iteration count identical and centroids <= 1e-5."""
import os

import numpy as np
import pytest
import torch

from anyloc_amd import synth
from oracle import cbuild, vlad_ref


@pytest.fixture(scope="module")
def co():
    return cbuild.load()


@pytest.mark.parametrize("tag", ["c2_n529_d1536_k32", "c5_n1369_d1024_k64"])
def test_c_vlad_against_reference_recordings(golden_dir, co, tag):
    g = np.load(os.path.join(golden_dir, f"vlad_{tag}.npz"))
    n_img, N, D, K = int(g["n_img"]), int(g["N"]), int(g["D"]), int(g["K"])
    x = synth.clustered_tokens(n_img, N, D, n_modes=K + 5, seed=int(g["seed"])).numpy()
    off = np.arange(n_img + 1, dtype=np.int64) * N
    v, lab, gap = co.vlad_hard(x.reshape(-1, D), off, g["centers"])
    want = g["labels"].astype(np.int64).reshape(-1)
    flips = lab != want
    assert (gap[flips] < 1e-6).all(), f"{flips.sum()} label flips, largest exact gap {gap[flips].max()}"
    clean = ~flips.reshape(n_img, N).any(axis=1)
    assert clean.any()
    assert np.abs(v[clean] - g["vlads"][clean]).max() <= 2e-6
    # tokens as passed need not be unit rows: labels from the raw rows, residuals from the re-normalised ones
    xr = x * g["scale"][:, :, None]
    vr, labr, gapr = co.vlad_hard(xr.reshape(-1, D), off, g["centers"])
    ok = ~(labr != want).reshape(n_img, N).any(axis=1)
    assert ((labr == want) | (gapr < 1e-6)).all() and ok.any()
    assert np.abs(vr[ok] - g["vlads_raw"][ok]).max() <= 2e-6
    # ragged packing incl. an empty image: rows of the packed call equal the per-image calls, the empty image is zero
    off2 = np.array([0, 0, 7, N, N + 1], dtype=np.int64)
    v2, _, _ = co.vlad_hard(x.reshape(-1, D)[:N + 1], off2, g["centers"])
    assert not v2[0].any()
    v1, _, _ = co.vlad_hard(x.reshape(-1, D)[7:N], np.array([0, N - 7], dtype=np.int64), g["centers"])
    assert np.array_equal(v2[2], v1[0])


def test_c_kmeans_against_reference_recording(golden_dir, co):
    g = np.load(os.path.join(golden_dir, "kmeans_n20000_d64_k16.npz"))
    x = synth.clustered_tokens(1, int(g["n"]), int(g["D"]), n_modes=int(g["K"]), seed=int(g["seed"]), noise=0.6)[0].numpy()
    xn = co.l2norm_rows(x)                                   # VLAD.fit normalises the rows first (utilities.py:782)
    assert np.abs(xn - torch.nn.functional.normalize(torch.from_numpy(x)).numpy()).max() <= 1e-7
    centers, iters = co.kmeans_fit(xn, int(g["K"]), g["init_idx"])
    assert iters == int(g["iters"])
    assert np.abs(centers - g["centers"]).max() <= 1e-5


def test_c_kmeans_empty_cluster_and_euclidean(co):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((400, 8)).astype(np.float32)
    x[:200] += 6.0
    # two identical initial rows: the later duplicate never wins an arg-max -> empty -> its centre becomes 0 (fpk: NaN -> 0)
    x[1] = x[0]
    c, it = co.kmeans_fit(x, 3, np.array([0, 1, 300]), mode="euclidean", max_iter=1)
    assert it == 1 and not c[1].any()
    km = vlad_ref.KMeans(3, mode="euclidean", max_iter=1)
    km.fit(torch.from_numpy(x), centroids=torch.from_numpy(x[[0, 1, 300]].copy()))
    assert np.abs(c - km.centroids.numpy()).max() <= 1e-5


def test_c_flat_search_against_reference_recordings(golden_dir, co):
    g = np.load(os.path.join(golden_dir, "config1_vits14_l9_value_k8.npz"))
    n_db = int(g["n_db"])
    vl = g["vlads"]
    top_k = list(range(1, 21))
    _, _, gt = synth.synthetic_places(n_db, int(g["n_qu"]), int(g["hw"]), int(g["hw"]), seed=int(g["images_seed"]))
    for metric, sfx in (("ip", ""), ("l2", "_l2")):
        # get_top_k_recall normalises both sides (utilities.py:436-437); the database through the flag
        d, i = co.flat_topk(co.l2norm_rows(vl[n_db:]), vl[:n_db], 20, metric, normalize_db=True)
        assert np.array_equal(i, g["top_idx" + sfx])
        assert np.abs(d - g["top_dist" + sfx]).max() <= 2e-6
        r = co.recalls(i, top_k, gt)
        assert [r[k] for k in top_k] == list(g["recalls" + sfx])


def test_c_flat_search_padding_and_ties(co):
    db = np.array([[1, 0], [0, 1], [1, 0], [0.5, 0.5]], dtype=np.float32)
    qu = np.array([[1, 0]], dtype=np.float32)
    d, i = co.flat_topk(qu, db, 6, "ip")
    assert i.tolist() == [[0, 2, 3, 1, -1, -1]]             # equal scores keep index order; k > ndb pads with -1
    assert d[0, :4].tolist() == [1.0, 1.0, 0.5, 0.0] and np.isneginf(d[0, 4:]).all()
    d, i = co.flat_topk(qu, db, 5, "l2")
    assert i.tolist() == [[0, 2, 3, 1, -1]] and np.isposinf(d[0, 4])
    assert d[0, :4].tolist() == [0.0, 0.0, 0.5, 2.0]


def test_c_labels_match_torch_restatement_on_random_rows(co):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3000, 96, generator=g) * torch.logspace(-3, 3, 3000)[:, None]
    c = torch.randn(17, 96, generator=g)
    for mode in ("cosine", "euclidean"):
        lab, gap = co.fpk_labels(x.numpy(), c.numpy(), mode)
        km = vlad_ref.KMeans(17, mode=mode)
        km.centroids = c
        want = km.predict(x).numpy()
        scale = 1.0 if mode == "cosine" else float((x * x).sum(1).max())
        assert (gap[lab != want] < 1e-5 * scale).all()
        assert (lab != want).mean() < 0.01


def test_c_dinov2_forward_against_reference_recordings(golden_dir, co):
    """The C restatement of the hub model's forward + the reference's facet hook (oracle_vit_facet) on the config-1 run:
    ViT-S/14, layer 9, `value` -- the tokens the REFERENCE's own __call__ recorded for images 0 and 31 -- and the other
    facets' probe projections; then a SwiGLU model (the ViT-g architecture at depth 2) against the torch restatement."""
    from oracle import dinov2_ref
    from oracle.make_golden import probe_vector
    g = np.load(os.path.join(golden_dir, "config1_vits14_l9_value_k8.npz"))
    name = str(g["model"])
    sd = synth.synthetic_state_dict(name, int(g["weights_seed"]))
    db, qu, _ = synth.synthetic_places(int(g["n_db"]), int(g["n_qu"]), int(g["hw"]), int(g["hw"]), seed=int(g["images_seed"]))
    imgs = torch.cat([db, qu])
    tok = co.vit_facet(name, sd, imgs[[0, 31]].numpy(), int(g["layer"]), str(g["facet"]))
    assert np.abs(tok[0] - g["tokens_img0"]).max() <= 1e-6 and np.abs(tok[1] - g["tokens_img31"]).max() <= 1e-6
    pv = probe_vector(384).numpy().astype(np.float64)
    for fname, kw in {"query": dict(layer=9, facet="query"), "key": dict(layer=9, facet="key"),
                      "token": dict(layer=9, facet="token"),
                      "value_cls_raw": dict(layer=9, facet="value", use_cls=True, norm_descs=False),
                      "token_l11": dict(layer=11, facet="token")}.items():
        out = co.vit_facet(name, sd, imgs[:1].numpy(), **kw)[0]
        assert tuple(out.shape) == tuple(g[f"facet_{fname}_shape"])
        want = g[f"facet_{fname}_proj"].astype(np.float64)
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(out.astype(np.float64) @ pv - want).max() <= 2e-5 * scale, fname
    gname = "dinov2_vitg14"
    sdg = synth.synthetic_state_dict(gname, 4, depth=2)
    model = dinov2_ref.DinoVisionTransformer(gname)
    model.blocks = model.blocks[:2]
    model.load_state_dict(sdg, strict=True)
    img = torch.randn(1, 3, 42, 70, generator=torch.Generator().manual_seed(9))
    for facet in ("value", "token"):
        want = dinov2_ref.extract_facet(model.eval(), img, 1, facet).numpy()
        got = co.vit_facet(gname, sdg, img.numpy(), 1, facet)
        assert np.abs(got - want).max() <= 1e-6, facet
