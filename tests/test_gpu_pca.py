"""Device PCA (anyloc_amd.pca: Gram / scatter matrix and projections on anyloc_gemm_nt) against the
reference's own implementation -- sklearn.decomposition.PCA(svd_solver='full'), which is what
utilities.py:561-564 calls -- on seeded matrices with a well-separated spectrum."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def decaying(n, f, seed, decay=0.9, rank=None):
    g = torch.Generator().manual_seed(seed)
    r = min(n, f) if rank is None else rank
    q1, _ = torch.linalg.qr(torch.randn(n, r, generator=g, dtype=torch.float64))
    q2, _ = torch.linalg.qr(torch.randn(f, r, generator=g, dtype=torch.float64))
    s = 10.0 * decay ** torch.arange(r, dtype=torch.float64)
    return ((q1 * s) @ q2.t() + 0.3 * torch.randn(1, f, generator=g, dtype=torch.float64)).float()


@pytest.mark.parametrize("shape,k,whiten", [((200, 3072), 32, False), ((200, 3072), 32, True),
                                            ((1000, 384), 64, False), ((1000, 384), 64, True), ((130, 131), 9, False)])
def test_pca_matches_sklearn_full_svd(shape, k, whiten):
    from sklearn.decomposition import PCA as SkPCA
    from anyloc_amd import pca
    x, y = decaying(*shape, seed=1, rank=100), decaying(33, shape[1], seed=2, rank=33)
    sk = SkPCA(k, svd_solver="full", whiten=whiten)
    want_tr = sk.fit_transform(x.double().numpy())          # float64 LAPACK: the reference method at full accuracy
    want_ts = sk.transform(y.double().numpy())
    ours = pca.PCA(k, whiten=whiten)
    got_tr, got_ts = ours.fit_transform(x.to(DEV)), ours.transform(y.to(DEV))
    assert got_tr.is_cuda and got_tr.shape == (shape[0], k) and got_ts.shape == (33, k)
    assert np.abs(ours.components_.cpu().numpy() - sk.components_).max() < 2e-5
    assert np.allclose(ours.explained_variance_.cpu().numpy(), sk.explained_variance_, rtol=2e-5)
    scale = np.abs(want_tr).max()
    # whitening divides the trailing axes' fp32 projections (abs error ~1e-7 |x|) by their tiny standard deviation
    tol = 5e-4 if whiten else 1e-4
    assert np.abs(got_tr.cpu().numpy() - want_tr).max() < tol * scale
    assert np.abs(got_ts.cpu().numpy() - want_ts).max() < tol * max(scale, np.abs(want_ts).max())
    # sklearn run in float32 (what the reference actually feeds it) is no closer to the float64 result
    sk32 = SkPCA(k, svd_solver="full", whiten=whiten)
    ref32 = sk32.fit_transform(x.numpy())
    assert np.abs(got_tr.cpu().numpy() - want_tr).max() < 4 * np.abs(ref32 - want_tr).max() + 1e-5 * scale
    # numpy in -> numpy out through the drop-in function
    import utilities
    a, b = utilities.reduce_pca(x.numpy(), y.numpy(), k, whitening=whiten)
    assert isinstance(a, np.ndarray) and np.abs(a - want_tr).max() < tol * scale and b.shape == (33, k)


def test_pca_all_fp32_kernel_mode_leading_axes():
    """precise=False: Gram matrix and back-projection on the fp32 MFMA kernel; the leading axes (variance within
    ~1e-3 of the largest) still agree with the float64 SVD."""
    from sklearn.decomposition import PCA as SkPCA
    from anyloc_amd import pca
    x = decaying(300, 2048, seed=3, decay=0.9, rank=100)
    sk = SkPCA(24, svd_solver="full").fit(x.double().numpy())
    ours = pca.PCA(24, precise=False).fit(x.to(DEV))
    assert np.abs(ours.components_.cpu().numpy() - sk.components_).max() < 1e-4
    assert np.allclose(ours.singular_values_.cpu().numpy(), sk.singular_values_, rtol=1e-4)


def test_pca_then_retrieval_keeps_ranking():
    """The use the reference makes of it (scripts/dino_v2_vlad.py:357-372): project db + queries, re-normalise,
    retrieve.  Data living in a 64-d subspace: a 64-d PCA keeps every inner product, so the top-k lists of the
    projected and the raw descriptors coincide."""
    from anyloc_amd import ops, pca
    g = torch.Generator().manual_seed(7)
    basis, _ = torch.linalg.qr(torch.randn(4096, 64, generator=g))
    coef = torch.randn(600, 64, generator=g) * (0.97 ** torch.arange(64))
    db = (coef @ basis.t()).to(DEV)
    qu = db[::10] + 0.05 * (torch.randn(60, 64, generator=g) @ basis.t()).to(DEV)
    p = pca.PCA(64).fit(db)
    mean = p.mean_
    raw_db, raw_qu = ops.l2norm_rows(db - mean), ops.l2norm_rows(qu - mean)
    red_db, red_qu = ops.l2norm_rows(p.transform(db)), ops.l2norm_rows(p.transform(qu))
    d0, i0 = ops.topk(raw_qu, raw_db, 5, "ip")
    d1, i1 = ops.topk(red_qu, red_db, 5, "ip")
    assert torch.equal(i0[:, 0], torch.arange(0, 600, 10, device=DEV))
    assert float((i0 != i1).float().mean()) < 0.01 and float((d0 - d1).abs().max()) < 1e-4


def test_joint_pca_project_matches_reference_script_expression():
    """scripts/joint_pca_project.py:62-101: one PCA fitted on the concatenated databases of several datasets, every
    database / query set projected with it and split back."""
    from sklearn.decomposition import PCA as SkPCA
    from anyloc_amd import pca
    dbs = [decaying(120, 1024, seed=11, rank=60), decaying(90, 1024, seed=12, rank=60)]
    qus = [decaying(20, 1024, seed=13, rank=20), decaying(31, 1024, seed=14, rank=31)]
    sk = SkPCA(n_components=48, whiten=False)
    want_db = sk.fit_transform(np.concatenate([d.double().numpy() for d in dbs]))
    want_qu = sk.transform(np.concatenate([q.double().numpy() for q in qus]))
    out_db, out_qu, fitted = pca.joint_pca_project(dbs, qus, 48, whiten=False)
    assert [tuple(o.shape) for o in out_db] == [(120, 48), (90, 48)] and [tuple(o.shape) for o in out_qu] == [(20, 48), (31, 48)]
    assert out_db[0].device.type == "cpu"                              # CPU tensors in -> CPU tensors out
    got_db, got_qu = torch.cat(out_db).numpy(), torch.cat(out_qu).numpy()
    scale = np.abs(want_db).max()
    assert np.abs(got_db - want_db).max() < 5e-4 * scale and np.abs(got_qu - want_qu).max() < 5e-4 * max(scale, np.abs(want_qu).max())
    a, b, _ = pca.joint_pca_project([d.numpy() for d in dbs], [q.numpy() for q in qus], 48, whiten=True)
    assert isinstance(a[1], np.ndarray) and a[1].shape == (90, 48)
    skw = SkPCA(n_components=48, whiten=True).fit(np.concatenate([d.double().numpy() for d in dbs]))
    want_w = skw.transform(np.concatenate([q.double().numpy() for q in qus]))
    # whitening divides the trailing axes' fp32 projections by a standard deviation ~100x below the leading one
    assert np.abs(np.concatenate(b) - want_w).max() < 5e-3 * np.abs(want_w).max()


def test_pca_u_based_sign_rule_and_rank_deficient_axes():
    """sign_convention='u' = svd_flip of the sklearn versions the reference pins (largest |entry| of every U column
    positive); n_components == n_samples leaves one direction without variance after centring -- it must come out as a
    zero axis (zero coordinates, also when whitening), not inf / NaN."""
    from anyloc_amd import pca
    x = decaying(40, 512, seed=21, rank=40)
    xc = (x - x.mean(0)).double()
    u, s, vt = torch.linalg.svd(xc, full_matrices=False)
    sign_u = torch.sign(u[u.abs().argmax(dim=0), torch.arange(u.shape[1])])          # u-based svd_flip
    want = (vt * sign_u[:, None])[:12]
    ours = pca.PCA(12, sign_convention="u").fit(x.to(DEV))
    assert float((ours.components_.cpu().double() - want).abs().max()) < 2e-5
    assert pca.PCA(12).sign_convention == pca.default_sign_convention()               # "auto": the installed sklearn's rule
    v_rule = pca.PCA(12, sign_convention="v").fit(x.to(DEV)).components_.cpu()
    assert float((v_rule.abs() - ours.components_.cpu().abs()).abs().max()) < 2e-5   # same axes up to sign
    full = pca.PCA(40, whiten=True).fit(x.to(DEV))                                     # rank 39 after centring
    z = full.transform(x.to(DEV))
    assert bool(torch.isfinite(z).all()) and bool(torch.isfinite(full.components_).all())
    assert float(z[:, -1].abs().max()) == 0.0 and float(full.components_[-1].abs().max()) == 0.0
    assert float((z[:, :39].var(dim=0, unbiased=True) - 1.0).abs().max()) < 1e-2      # whitened: unit variance


@pytest.mark.parametrize("n,f", [(70, 130), (130, 70), (64, 64), (1, 5), (257, 1031), (260, 1032), (1032, 132)])
def test_pca_f64_products_vs_float64(n, f):
    """csrc/pca_f64.hip (the float64 matrix-core products of the fit) against the same products in torch float64 of the
    same centred data: Gram, scatter (both mirrored from the upper tiles) and the back-projection, ragged shapes."""
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(n * 1000 + f)
    x = (torch.randn(n, f, generator=g) * 3 + 40.0).to(DEV)            # a large common offset: centring must be in float64
    mean = x.mean(dim=0, dtype=torch.float64)
    xw = x.double() - mean
    for side, want in ((0, xw @ xw.t()), (1, xw.t() @ xw)):
        got = ops.pca_gram_f64(x, mean, side)
        assert got.dtype == torch.float64 and got.shape == want.shape
        assert torch.equal(got, got.t())                                  # mirrored, not recomputed
        assert float((got - want).abs().max()) <= 1e-13 * float(want.abs().max()) * max(n, f) ** 0.5
    k = min(n, 7)
    vec = torch.linalg.qr(torch.randn(n, n, generator=g, dtype=torch.float64))[0].to(DEV)
    got = ops.pca_axes_f64(vec, k, x, mean)
    want = vec[:, :k].t() @ xw
    assert got.shape == (k, f) and float((got - want).abs().max()) <= 1e-13 * float(want.abs().max() + 1) * n ** 0.5
    # eigenvectors are read where they lie: a view of the first k columns, and column-major storage (what eigh returns)
    assert torch.equal(ops.pca_axes_f64(vec[:, :k], k, x, mean), got)
    assert torch.equal(ops.pca_axes_f64(vec.t().contiguous().t(), k, x, mean), got)
    with pytest.raises(ValueError):
        ops.pca_gram_f64(x, mean[:-1], 0) if f > 1 else ops.pca_axes_f64(vec[:-1], k, x, mean)


def test_pca_fit_makes_no_float64_copy_of_the_data():
    """The precise fit reads the fp32 descriptors where they lie: no float64 copy (2 x the data, what the torch.matmul
    route needed) and no centred fp32 copy (1 x), at a shape where the symmetric matrix is small."""
    from anyloc_amd import pca
    n, f = 1024, 49152
    x = decaying(n, f, seed=5, rank=64).to(DEV)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    pca.PCA(16).fit(x)
    torch.cuda.synchronize()
    extra = torch.cuda.max_memory_allocated() - base
    assert extra < 0.5 * x.numel() * 4, (extra, x.numel() * 4)
