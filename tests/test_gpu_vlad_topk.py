"""Parity of the HIP VLAD / k-means / top-k path (through the C ABI) against the golden
vectors recorded from the reference and against the CPU oracle on seeded inputs."""
import os

import numpy as np
import pytest
import torch

from anyloc_amd import synth
from oracle import faiss_flat, vlad_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
VLAD_RTOL = 1e-5      # north_star: VLAD descriptors within 1e-5 relative (L2-relative, fp32)


def l2rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def check_labels(lab, lab_ref, x, centers, gap_tol=1e-6):
    """Cluster ids bit-exact; a differing id is tolerated only if the oracle's top-2 cosine
    gap for that token is below fp32 resolution (SURVEY section 7 'Argmax bit-exactness')."""
    lab, lab_ref = torch.as_tensor(lab).cpu().long(), torch.as_tensor(lab_ref).cpu().long()
    bad = (lab != lab_ref).nonzero().flatten()
    if len(bad) == 0:
        return 0
    sc = vlad_ref.fpk_cosine_scores(x.reshape(-1, x.shape[-1])[bad].cpu(), centers.cpu())
    top2 = sc.topk(2, dim=1)[0]
    gap = top2[:, 0] - top2[:, 1]
    assert float(gap.max()) < gap_tol, f"{len(bad)} label flips, largest oracle gap {float(gap.max()):.3e}"
    return len(bad)


def check_labels_sim(lab, lab_ref, sim, gap_tol=1e-6):
    """The same rule for any similarity matrix of the oracle (``sim`` [n, K], larger = closer): a differing id must sit
    at an oracle top-2 gap below ``gap_tol`` (relative to the winning score where that exceeds 1)."""
    lab, lab_ref = torch.as_tensor(lab).cpu().long(), torch.as_tensor(lab_ref).cpu().long()
    bad = (lab != lab_ref).nonzero().flatten()
    if len(bad) == 0:
        return 0
    top2 = sim[bad].topk(2, dim=1)[0]
    gap = (top2[:, 0] - top2[:, 1]) / top2[:, 0].abs().clamp_min(1.0)
    assert float(gap.max()) < gap_tol, f"{len(bad)} label flips, largest oracle gap {float(gap.max()):.3e}"
    # ... and the id the kernel chose must be the oracle's runner-up, not some third cluster
    assert torch.equal(lab[bad], sim[bad].topk(2, dim=1)[1][:, 1]), "a flipped id is not the oracle's runner-up"
    return len(bad)


@pytest.mark.parametrize("tag", ["c2_n529_d1536_k32", "c5_n1369_d1024_k64"])
def test_vlad_hard_golden(golden_dir, tag):
    from anyloc_amd import ops
    g = np.load(os.path.join(golden_dir, f"vlad_{tag}.npz"))
    x = synth.clustered_tokens(int(g["n_img"]), int(g["N"]), int(g["D"]), n_modes=int(g["K"]) + 5,
                               seed=int(g["seed"]))
    centers = torch.from_numpy(g["centers"])
    out, lab = ops.vlad(x.to(DEV), centers.to(DEV), return_labels=True)
    assert check_labels(lab.reshape(x.shape[0], -1), g["labels"].astype(np.int64), x, centers) == 0
    for i in range(x.shape[0]):
        assert l2rel(out[i], g["vlads"][i]) < VLAD_RTOL
    xr = x * torch.from_numpy(g["scale"])[:, :, None]
    out_r = ops.vlad(xr.to(DEV), centers.to(DEV))
    for i in range(x.shape[0]):
        assert l2rel(out_r[i], g["vlads_raw"][i]) < VLAD_RTOL
    # unused clusters are exact zero blocks (reference utilities.py:840,854-861)
    K, D = centers.shape
    used = set(g["labels"][0].tolist())
    for k in range(K):
        if k not in used:
            assert float(out[0, k * D:(k + 1) * D].abs().max()) == 0.0


@pytest.mark.parametrize("K,D,N", [(8, 384, 256), (32, 1536, 529), (64, 1024, 300), (5, 64, 77), (128, 128, 500), (200, 256, 700)])
def test_vlad_hard_vs_oracle_flags_and_ragged(K, D, N):
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(K * D + N)
    x = synth.clustered_tokens(3, N, D, n_modes=K, seed=K + N) * (0.5 + torch.rand(3, N, 1, generator=g))
    centers = 0.7 * synth.clustered_tokens(1, K, D, n_modes=K, seed=K + N)[0] + 0.01 * torch.randn(K, D, generator=g)
    for norm_descs, intra in ((True, True), (False, True), (True, False)):
        out, lab = ops.vlad(x.to(DEV), centers.to(DEV), norm_descs=norm_descs, intra_norm=intra,
                            return_labels=True)
        for i in range(2):
            v, l = vlad_ref.vlad_hard(x[i], centers, norm_descs, intra)
            lab_i = lab[i * N:(i + 1) * N].cpu()
            if check_labels(lab_i, l, x[i], centers):
                # a token sat on a cosine tie (< 1e-6): score the descriptor under the same assignment
                v = vlad_ref.vlad_hard(x[i], centers, norm_descs, intra, labels=lab_i)[0]
            assert l2rel(out[i], v) < VLAD_RTOL
    # ragged list with an empty image and a single-token image
    parts = [x[0, :N // 2], x[1, :0], x[2, :1], x[2]]
    out = ops.vlad([p.to(DEV) for p in parts], centers.to(DEV))
    assert out.shape == (4, K * D)
    assert float(out[1].abs().max()) == 0.0
    for i in (0, 2, 3):
        assert l2rel(out[i], vlad_ref.vlad_hard(parts[i], centers)[0]) < VLAD_RTOL


@pytest.mark.parametrize("opts", ["vlad_parts=1", "vlad_parts=3", "vlad_parts=8", "vlad_parts=40", "", "vlad_two_pass=1",
                                  "vlad_fused_v=1,kmeans_fused_v=1", "vlad_fused_v=3,kmeans_fused_v=3,vlad_parts=2",
                                  "vlad_fused_v=4,kmeans_fused_v=4,vlad_parts=5"])
def test_vlad_fused_parts(opts):
    """The fused VLAD kernels with 1 / 3 / 8 / 40 (more parts than tiles) workgroups per image, their own choice, every
    kernel version (options *_fused_v: the exact-score kernel 1, the fp16-screening kernel with 4 and 8 waves) and the
    two-pass path: each against the oracle, bitwise reproducible run to run; plus one k-means step on close-call inputs
    (tests/_vlad_parts_job.py, a fresh process configured through ANYLOC_OPTIONS)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), ANYLOC_OPTIONS=opts)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_vlad_parts_job.py")], env=e, cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok worst=" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("K,D,N", [(32, 1536, 529), (16, 384, 300), (40, 512, 257)])
def test_vlad_hard_euclidean_assignment(K, D, N):
    """VLAD(dist_mode='euclidean'): labels = kmeans.predict(tokens) with fpk's euclidean similarity on the tokens as
    passed (reference utilities.py:849 with self.mode = 'euclidean'), on the fused kernel (D=1536/384) and on the
    two-pass path (D=512).  Tokens are NOT unit norm, so the euclidean and the cosine assignment differ."""
    import utilities
    from anyloc_amd import ops
    from oracle.fpk_kmeans import KMeans as RefKM
    g = torch.Generator().manual_seed(K + D + N)
    x = synth.clustered_tokens(3, N, D, n_modes=K, seed=K + N) * (0.3 + 1.4 * torch.rand(3, N, 1, generator=g))
    centers = synth.clustered_tokens(1, K, D, n_modes=K, seed=K + N)[0] * (0.4 + torch.rand(K, 1, generator=g))
    out, lab = ops.vlad(x.to(DEV), centers.to(DEV), return_labels=True, dist_mode="euclidean")
    lab = lab.cpu().reshape(3, N)
    n_diff_metric = 0
    for i in range(3):
        sim = RefKM.euc_sim(x[i], centers)
        ref_lab = sim.max(dim=-1)[1]
        bad = (lab[i] != ref_lab).nonzero().flatten()
        if len(bad):      # tolerated only at fp32 ties of the oracle's own similarity (values are O(1))
            top2 = sim[bad].topk(2, dim=1)[0]
            assert float((top2[:, 0] - top2[:, 1]).max()) < 2e-6
        v = vlad_ref.vlad_hard(x[i], centers, labels=lab[i])[0]
        assert l2rel(out[i], v) < VLAD_RTOL
        n_diff_metric += int((ref_lab != vlad_ref.hard_labels(x[i], centers)).sum())
    assert n_diff_metric > 0                               # the case really separates the two metrics
    # through the class surface: VLAD(dist_mode="euclidean").generate == the same labels
    vl = utilities.VLAD(K, D, dist_mode="euclidean", cache_dir=None)
    vl.c_centers = centers
    vl.kmeans = utilities.KMeans(K, mode="euclidean")
    vl.kmeans.centroids = centers
    assert l2rel(vl.generate(x[0]), out[0]) < 1e-6
    assert torch.equal(vl.kmeans.predict(x[0]), lab[0])


def test_vlad_soft_vs_oracle():
    from anyloc_amd import ops
    K, D, N = 8, 384, 256
    x = synth.clustered_tokens(2, N, D, n_modes=K, seed=3)
    g = torch.Generator().manual_seed(9)
    centers = 0.7 * synth.clustered_tokens(1, K, D, n_modes=K, seed=3)[0] + 0.01 * torch.randn(K, D, generator=g)
    for temp in (1.0, 7.5):
        out = ops.vlad(x.to(DEV), centers.to(DEV), mode="soft", soft_temp=temp)
        for i in range(2):
            ref = vlad_ref.vlad_soft(x[i], centers, temp)[0]
            assert l2rel(out[i], ref) < VLAD_RTOL, l2rel(out[i], ref)


def test_kmeans_golden_and_oracle(golden_dir):
    from anyloc_amd import kmeans as hk
    g = np.load(os.path.join(golden_dir, "kmeans_n20000_d64_k16.npz"))
    x = synth.clustered_tokens(1, int(g["n"]), int(g["D"]), n_modes=int(g["K"]), seed=int(g["seed"]), noise=0.6)[0]
    km = hk.KMeans(int(g["K"]), mode="cosine")
    xn = torch.nn.functional.normalize(x)
    labels = km.fit_predict(xn.to(DEV), centroids=xn[torch.from_numpy(g["init_idx"])].to(DEV))
    assert km.n_iter_ == int(g["iters"])
    assert l2rel(km.centroids, g["centers"]) < 1e-5
    # fit_predict returns the assignment computed BEFORE the last update (fpk semantics)
    from oracle.fpk_kmeans import KMeans as RefKM
    ref = RefKM(int(g["K"]), mode="cosine")
    init = xn[torch.from_numpy(g["init_idx"])]
    ref_lab = ref.fit_predict(xn, centroids=init.clone())
    # ... i.e. against the centroids after n_iter - 1 updates; a differing id is accepted only at an oracle top-2 gap
    # below fp32 resolution, and only as the oracle's runner-up
    if ref.n_iter_ > 1:
        prev = RefKM(int(g["K"]), mode="cosine", max_iter=ref.n_iter_ - 1)
        prev.fit(xn, centroids=init.clone())
        c_prev = prev.centroids
    else:
        c_prev = init
    check_labels_sim(labels, ref_lab, RefKM.cos_sim(xn, c_prev))
    check_labels_sim(km.predict(xn.to(DEV)), ref.predict(xn), RefKM.cos_sim(xn, ref.centroids))
    # euclidean mode + empty clusters vs the oracle, one step
    from anyloc_amd import ops
    c = torch.cat([x[:5], 50 + torch.zeros(2, x.shape[1])])      # two centres nobody picks
    sums, counts, lab = ops.kmeans_step(x.to(DEV), c.to(DEV), "euclidean", True)
    sim_e = RefKM.euc_sim(x, c)
    ref_lab = sim_e.max(dim=-1)[1]
    check_labels_sim(lab, ref_lab, sim_e)
    onehot = (lab.cpu()[None] == torch.arange(7)[:, None]).float()
    assert torch.equal(counts.cpu(), onehot.sum(-1))
    assert l2rel(sums, onehot.double() @ x.double()) < 1e-6
    assert float(counts[5:].sum()) == 0.0


@pytest.mark.parametrize("n,D,K,mode", [(5003, 384, 16, "cosine"), (3000, 1536, 32, "cosine"),
                                        (2500, 768, 7, "euclidean"), (17, 1024, 3, "cosine")])
def test_kmeans_step_fused_path_vs_oracle(n, D, K, mode):
    """Shapes served by the single-pass fused kernel (D in {384,768,1024,1536}, K <= 32): one
    assign + accumulate step against the fast-pytorch-kmeans rule, incl. ragged last tile / chunk."""
    from anyloc_amd import ops
    from oracle.fpk_kmeans import KMeans as RefKM
    x = synth.clustered_tokens(1, n, D, n_modes=max(K - 1, 2), seed=n + K, noise=0.5)[0]
    g = torch.Generator().manual_seed(K)
    x = x * (0.5 + torch.rand(n, 1, generator=g))
    c = x[torch.randperm(n, generator=g)[:K]].clone() + 0.01 * torch.randn(K, D, generator=g)
    sums, counts, lab = ops.kmeans_step(x.to(DEV), c.to(DEV), mode, True)
    sim = RefKM.cos_sim(x, c) if mode == "cosine" else RefKM.euc_sim(x, c)
    ref_lab = sim.max(dim=-1)[1]
    check_labels_sim(lab, ref_lab, sim)
    lab_c = lab.cpu()
    onehot = (lab_c[None] == torch.arange(K)[:, None]).double()
    assert torch.equal(counts.cpu(), onehot.sum(-1).float())
    assert l2rel(sums, onehot @ x.double()) < 1e-6


def test_vlad_fit_surface_matches_reference_semantics(golden_dir, capsys):
    """utilities.VLAD.fit -> generate_multi on CPU tensors (the reference's calling convention)."""
    import utilities
    g = np.load(os.path.join(golden_dir, "kmeans_n20000_d64_k16.npz"))
    x = synth.clustered_tokens(1, int(g["n"]), int(g["D"]), n_modes=int(g["K"]), seed=int(g["seed"]), noise=0.6)[0]
    utilities.seed_everything(42)
    v = utilities.VLAD(int(g["K"]), None, cache_dir=None)
    v.fit(x)                                   # draws its init rows from NumPy's global RNG (seed 42)
    assert v.desc_dim == int(g["D"]) and v.c_centers.device.type == "cpu"
    assert v.kmeans.n_iter_ == int(g["iters"])
    assert l2rel(v.c_centers, g["centers"]) < 1e-5
    out = v.generate_multi(x[:600].reshape(3, 200, -1))
    assert out.device.type == "cpu" and out.shape == (3, int(g["K"]) * int(g["D"]))
    for i in range(3):
        ref = vlad_ref.vlad_hard(x[i * 200:(i + 1) * 200], torch.from_numpy(g["centers"]))[0]
        assert l2rel(out[i], ref) < 2e-5
    single = v.generate(x[:200])
    assert torch.equal(single, out[0])
    assert torch.equal(v.kmeans.predict(x[:50]), vlad_ref.hard_labels(x[:50], v.c_centers))


@pytest.mark.parametrize("nq,ndb,dim,k,metric", [(9, 103, 64, 12, "ip"), (9, 103, 64, 12, "l2"),
                                                  (33, 5000, 128, 20, "ip"), (4, 7, 32, 20, "ip"),
                                                  (4, 7, 32, 20, "l2"), (3, 40000, 64, 5, "ip"),
                                                  (17, 9000, 3072, 20, "ip")])
def test_topk_vs_oracle(nq, ndb, dim, k, metric):
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(nq + ndb + dim)
    db = torch.nn.functional.normalize(torch.randn(ndb, dim, generator=g))
    qu = torch.nn.functional.normalize(torch.randn(nq, dim, generator=g))
    if ndb > 50:
        db[40] = db[7]                      # exact duplicate: tie -> lower index first
        qu[0] = db[7]
    d, i = ops.topk(qu.to(DEV), db.to(DEV), k, metric)
    d_ref, i_ref = faiss_flat.flat_search(qu, db, k, metric)
    d, i = d.cpu(), i.cpu()
    kk = min(k, ndb)
    assert torch.equal(i[:, kk:], i_ref[:, kk:])                # -1 padding when k > ndb
    mism = (i[:, :kk] != i_ref[:, :kk])
    if mism.any():
        # only near-ties may swap: the reference distances at the swapped ranks must agree to fp32 noise
        gap = (d_ref[:, :kk][mism] - d[:, :kk][mism]).abs().max()
        assert float(gap) < 1e-5, float(gap)
    np.testing.assert_allclose(d[:, :kk].numpy(), d_ref[:, :kk].numpy(), rtol=0, atol=2e-6 if metric == "ip" else 1e-5)
    if ndb > 50:
        assert i[0, 0] == 7 and i[0, 1] == 40
    d2, i2 = ops.topk(qu.to(DEV), db.to(DEV), k, metric, index_base=1000)
    assert torch.equal(torch.where(i2.cpu() >= 0, i2.cpu() - 1000, i2.cpu()), i)


@pytest.mark.parametrize("nq,ndb,dim,k,metric", [(61, 3000, 4096, 20, "ip"), (61, 3000, 4096, 20, "l2"), (64, 700, 8192, 7, "ip"),
                                                  (1, 130, 49152, 20, "ip"), (5, 33000, 4096, 20, "ip"),
                                                  (200, 2000, 4096, 10, "ip"), (9, 103, 64, 12, "l2")])
def test_topk_normalize_db_vs_oracle(nq, ndb, dim, k, metric):
    """ANYLOC_TOPK_NORMALIZE_DB: the database is passed RAW and F.normalize(db) (reference utilities.py:436) is applied to
    the scores.  Covers the few-query split-K path (<= 64 queries, dim >= 4096: database rows as the GEMM's M operand,
    row norms from the same pass, several panels) and the panel-GEMM path, against normalise-then-flat-search."""
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(nq + ndb + dim)
    db = torch.randn(ndb, dim, generator=g) * (0.2 + 3.0 * torch.rand(ndb, 1, generator=g))   # norms all over the place
    qu = torch.nn.functional.normalize(torch.randn(nq, dim, generator=g))
    db[11] = 0.0                                        # a zero row stays zero under F.normalize
    db[40] = 2.5 * db[7]                                # same direction, different norm: an exact tie after normalising?
    qu[0] = torch.nn.functional.normalize(db[7], dim=0)
    d, i = ops.topk(qu.to(DEV), db.to(DEV), k, metric, normalize_db=True)
    dbn = torch.nn.functional.normalize(db)
    d_ref, i_ref = faiss_flat.flat_search(qu, dbn, k, metric)
    d64 = qu.double() @ torch.nn.functional.normalize(db.double()).t()
    if metric == "l2":
        d64 = 2.0 - 2.0 * d64
        d64[:, 11] = 1.0                                # |q|^2 + 0 - 0 for the zero row
    d, i = d.cpu(), i.cpu()
    kk = min(k, ndb)
    got64 = torch.gather(d64, 1, i[:, :kk])
    np.testing.assert_allclose(d[:, :kk].double().numpy(), got64.numpy(), atol=3e-6)      # each distance is the true one
    mism = i[:, :kk] != i_ref[:, :kk]
    if mism.any():                                      # only near-ties may swap
        assert float((d_ref[:, :kk][mism] - d[:, :kk][mism]).abs().max()) < 1e-5
    assert {int(i[0, 0]), int(i[0, 1])} == {7, 40}      # the two rows of the query's own direction come first
    assert bool((d[:, :kk - 1] >= d[:, 1:kk]).all()) if metric == "ip" else bool((d[:, :kk - 1] <= d[:, 1:kk]).all())
    # identical to normalising first and searching without the flag, up to fp32 noise
    d2, i2 = ops.topk(qu.to(DEV), dbn.to(DEV), k, metric)
    assert float((d2.cpu()[:, :kk] - d[:, :kk]).abs().max()) < 5e-6
    assert float((i2.cpu()[:, :kk] != i[:, :kk]).float().mean()) < 0.02


@pytest.mark.parametrize("order", ["ascending", "descending", "constant", "blocks"])
@pytest.mark.parametrize("k", [1, 20, 300])
def test_topk_merge_adversarial_orders(order, k):
    """The merge kernel keeps only the columns that beat the running k-th entry (csrc/topk.hip): score rows that rise
    monotonically (every tile beats the threshold: one selection per tile), fall, are all equal (every entry ties: lower
    index first) or come in equal-valued blocks must give exactly the flat search's lists -- over several panels."""
    from anyloc_amd import ops
    ndb, dim = 70001, 8                                   # three 32768-row panels, the last one ragged
    u = torch.zeros(dim)
    u[0] = 1.0
    j = torch.arange(ndb, dtype=torch.float32)
    if order == "ascending":
        s = (j + 1.0) / ndb
    elif order == "descending":
        s = (ndb - j) / ndb
    elif order == "constant":
        s = torch.full((ndb,), 0.5)
    else:
        s = torch.floor(j / 97.0) % 13 / 13.0 + 0.25      # blocks of 97 equal scores, 13 levels, repeating
    db = s[:, None] * u[None, :]
    qu = torch.stack([u, 2.0 * u, -u])                    # -u reverses the order
    for metric in ("ip", "l2"):
        d, i = ops.topk(qu.to(DEV), db.to(DEV), k, metric)
        d_ref, i_ref = faiss_flat.flat_search(qu, db, k, metric)
        d, i = d.cpu(), i.cpu()
        np.testing.assert_allclose(d.numpy(), d_ref.numpy(), rtol=0, atol=1e-6)
        if metric == "ip":                                # one exact product per score: the lists must be identical
            assert torch.equal(i, i_ref), (order, k, metric)
        else:                                             # |q|^2 + |d|^2 - 2 q.d rounds: only exact-distance ties may swap
            mism = i != i_ref
            assert not mism.any() or float((d_ref[mism] - d[mism]).abs().max()) < 1e-6
            for r in range(i.shape[0]):                   # ... and the GPU's own list is ordered (distance, then index)
                dd, ii = d[r], i[r]
                assert bool(((dd[:-1] < dd[1:]) | ((dd[:-1] == dd[1:]) & (ii[:-1] < ii[1:]))).all())


def test_get_top_k_recall_surface():
    import utilities
    g = torch.Generator().manual_seed(5)
    db = torch.randn(60, 96, generator=g)
    qu = db[:10] + 0.3 * torch.randn(10, 96, generator=g)
    gt = np.empty(10, dtype=object)
    for q in range(10):
        gt[q] = np.array([q])
    top_k = [1, 5, 10]
    for method in ("cosine", "l2"):
        d, i, r = utilities.get_top_k_recall(top_k, db, qu, gt, method=method)
        d0, i0, r0 = vlad_ref.top_k_recall(top_k, db, qu, gt, method=method)
        assert d.device.type == "cpu" and d.shape == (10, 10) and i.dtype == torch.int64
        assert torch.equal(i, i0) and r == r0
        np.testing.assert_allclose(d.numpy(), d0.numpy(), atol=1e-5)
    d, i, r = utilities.get_top_k_recall([1], db, qu[0], gt)       # 1-D query (utilities.py:433-434)
    assert i.shape == (1, 1)
    with pytest.raises(NotImplementedError):
        utilities.get_top_k_recall([1], db, qu, gt, method="manhattan")


def _vlad_f64(x, centers, labels):
    """The reference expression (utilities.py:959-962, :854-861, :889) evaluated in float64 under a given assignment."""
    K, D = centers.shape
    xh = torch.nn.functional.normalize(x.double())
    out = torch.zeros(K, D, dtype=torch.float64)
    out.index_add_(0, labels, xh - centers.double()[labels])
    out = torch.nn.functional.normalize(out, dim=1)
    return torch.nn.functional.normalize(out.reshape(-1), dim=0)


@pytest.mark.parametrize("D,case", [(1536, "random"), (1536, "one_cluster"), (1536, "outlier"), (1024, "random"),
                                    (768, "one_cluster"), (384, "random")])
def test_vlad_tight_clusters(D, case):
    """Tight clusters -- unit tokens 1e-2 away from a unit-norm centre -- through the fused kernel.  The stress amplifies the
    one rounding every fp32 implementation shares (x^ = x / ||x||, good to ~6e-8: ~6e-6 of a 1e-2 residual), so the yardstick
    is the reference's own fp32 arithmetic (the oracle): the kernel must be as close to a float64 evaluation as the oracle is
    (factor 3 + 1e-6) and within 2e-5 of the oracle.  (A plain sum x^ - n_k c_k -- k-means mode's loop with the centre
    subtracted once -- would be off by 4e-5 ... 7e-4 here: the reason the kernel subtracts the centre per token.)  Few images
    (several workgroups per image, partial sums handed over) and many (one workgroup per image); ids identical."""
    from anyloc_amd import ops
    K, N = 32, 529
    g = torch.Generator().manual_seed(D + len(case))
    c = torch.nn.functional.normalize(torch.randn(K, D, generator=g))
    if case == "outlier":
        c[:, 7] += 0.5                                     # a channel that is large in every centre
        c[3, 100] = -0.9
        c = torch.nn.functional.normalize(c)
    for n_img in (3, 140):
        lab = torch.full((n_img, N), 5, dtype=torch.long) if case == "one_cluster" else torch.randint(0, K, (n_img, N), generator=g)
        x = c[lab] + (1e-2 / D ** 0.5) * torch.randn(n_img, N, D, generator=g)
        x = x * (0.5 + 1.5 * torch.rand(n_img, N, 1, generator=g))           # raw tokens: the kernel normalises
        out, lab_g = ops.vlad(x.to(DEV), c.to(DEV), return_labels=True)
        assert torch.equal(lab_g.cpu().reshape(n_img, N), lab)
        worst = 0.0
        for i in (0, n_img // 2, n_img - 1):
            v32 = vlad_ref.vlad_hard(x[i], c)[0]
            v64 = _vlad_f64(x[i], c, lab[i])
            e_or = float((v32.double() - v64).norm())
            e_k = float((out[i].cpu().double() - v64).norm())
            assert e_k <= 3.0 * e_or + 1e-6, (case, D, n_img, i, e_k, e_or)
            assert l2rel(out[i], v32) <= 2e-5, (case, D, n_img, i, l2rel(out[i], v32))
            worst = max(worst, e_k / max(e_or, 1e-30))
        print(f"[tight {case} D={D} n_img={n_img}] kernel / oracle distance to float64: {worst:.2f}")


@pytest.mark.parametrize("D,N", [(1536, 529), (1024, 257)])
def test_vlad_reproducible_under_load(D, N):
    """300 images through the fused launch, 10 times: every run bitwise equal to the first, and the first within 1e-5 of the
    two-pass path wherever the two paths agree on the cluster ids.  Round 5 met a structure of the gather (a branch around
    the register-indexed adds, a fused multiply-add for the residual) whose results differed run to run on 1e-3 ... 1e-1 of the
    images at exactly these sizes while every small-batch test passed (profiles/r05_vlad_stress_bisect.log): the chip has to be
    full for it to show."""
    from anyloc_amd import ops
    K, n_img = 32, 300
    c = 0.8 * synth.clustered_tokens(1, K, D, n_modes=K, seed=3)[0].to(DEV)
    toks = synth.clustered_tokens(n_img, N, D, n_modes=K, seed=11, noise=0.6).to(DEV)
    with ops.options(vlad_two_pass=1):
        ref, lab_ref = ops.vlad(toks, c, return_labels=True)
    first, lab = ops.vlad(toks, c, return_labels=True)
    first = first.clone()
    same = (lab.reshape(n_img, N) == lab_ref.reshape(n_img, N)).all(dim=1)
    assert float(same.float().mean()) > 0.95
    rel = ((first - ref).norm(dim=1) / ref.norm(dim=1))[same]
    assert float(rel.max()) < 1e-5, float(rel.max())
    for rep in range(10):
        again = ops.vlad(toks, c)
        bad = (again != first).any(dim=1)
        assert not bool(bad.any()), (rep, int(bad.sum()), "image results differ from the first run")

