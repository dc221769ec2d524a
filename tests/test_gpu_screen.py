"""Screened retrieval (``pytest -m gpu``; csrc/scores_screen.hip, option ``topk_screen``, ABI 9
``anyloc_topk_search_index_rows``): score panels on the leading fp16 planes alone under a proven bound, exact re-scoring of the
rows the bound cannot rule out.  The lists must be those of the exact search -- checked against a float64 flat search on the
device (ties -> lower index; a differing index only where the two float64 scores are closer than 3e-6) and against the
unscreened kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _exact64(qu, db, k, metric):
    """float64 flat search over F.normalize(db) on the device: (values, indices), ties -> lower index."""
    q = qu.double()
    d = torch.nn.functional.normalize(db.double())
    s = q @ d.t()
    if metric == "l2":
        s = -((q * q).sum(1, keepdim=True) + (d * d).sum(1)[None, :] - 2.0 * s)
    order = torch.sort(s, dim=1, descending=True, stable=True)
    v, i = order.values[:, :k], order.indices[:, :k]
    return (-v if metric == "l2" else v), i, s


def _check(d, i, qu, db, k, metric, tag):
    v64, i64, s = _exact64(qu, db, k, metric)
    kk = min(k, db.shape[0])
    assert bool((i[:, kk:] == -1).all()), tag
    got64 = torch.gather(s, 1, i[:, :kk])
    got64 = -got64 if metric == "l2" else got64
    tol = 3e-6 if metric == "ip" else 1e-5                # (the bars of tests/test_gpu_vlad_topk.py)
    assert float((d[:, :kk].double() - got64).abs().max()) <= tol, (tag, float((d[:, :kk].double() - got64).abs().max()))
    mism = i[:, :kk] != i64[:, :kk]
    if bool(mism.any()):                                  # only float64 near-ties may swap
        assert float((got64[mism] - v64[:, :kk][mism]).abs().max()) <= tol, tag
    return int(mism.sum())


def _data(nq, ndb, dim, seed, planted=True):
    g = torch.Generator(device=DEV).manual_seed(seed)
    db = torch.randn(ndb, dim, generator=g, device=DEV) * (0.3 + 2.0 * torch.rand(ndb, 1, generator=g, device=DEV))
    qu = torch.nn.functional.normalize(torch.randn(nq, dim, generator=g, device=DEV))
    if planted and ndb >= 64:
        # every query has a handful of true neighbours at graded distances, some closer to each other than the screening bound
        for j in range(6):
            rows = torch.randint(0, ndb, (nq,), generator=g, device=DEV)
            noise = torch.nn.functional.normalize(torch.randn(nq, dim, generator=g, device=DEV))
            db[rows] = (qu + (0.02 + 0.0004 * j) * noise) * (0.5 + j)
    return qu, db


@pytest.mark.parametrize("nq,ndb,dim,k,metric", [(600, 20000, 4096, 20, "ip"), (600, 20000, 4096, 20, "l2"), (257, 9000, 1024, 5, "ip"),
                                                  (300, 140000, 512, 10, "ip"), (130, 300, 2048, 20, "ip"), (70, 12, 256, 20, "l2"),
                                                  (1000, 10000, 49152, 20, "ip")])
def test_screened_search_gives_the_exact_lists(nq, ndb, dim, k, metric):
    """Option topk_screen = 1 against the float64 flat search and against the unscreened panels: every shape class -- several
    panels, a ragged last panel, two column ranges (140 000 rows), fewer rows than k, the bench's 1 000 x 10 000 x 49 152."""
    from anyloc_amd import _lib, ops
    qu, db = _data(nq, ndb, dim, nq + ndb + dim)
    with ops.options(topk_screen=1, topk_h3=1):
        ops.profile_enable(True); ops.profile_reset()
        d, i = ops.topk(qu, db, k, metric, normalize_db=True)
        torch.cuda.synchronize()
        prof = ops.profile_dump()
        ops.profile_enable(False)
        assert "topk_screen_gemm" in prof and "topk_scores_gemm" not in prof, sorted(prof)     # screened, and no fallback
        d_again, i_again = ops.topk(qu, db, k, metric, normalize_db=True)
        assert torch.equal(d, d_again) and torch.equal(i, i_again)                              # deterministic
        d_b, i_b = ops.topk(qu, db, k, metric, normalize_db=True, index_base=5000)
        assert torch.equal(torch.where(i_b >= 0, i_b - 5000, i_b), i) and torch.equal(d_b, d)
    with ops.options(topk_screen=0, topk_h3=1):
        d0, i0 = ops.topk(qu, db, k, metric, normalize_db=True)
    swaps = _check(d, i, qu, db, k, metric, "screened")
    swaps0 = _check(d0, i0, qu, db, k, metric, "unscreened")
    kk = min(k, ndb)
    assert float((d[:, :kk] - d0[:, :kk]).abs().max()) <= (3e-6 if metric == "ip" else 1e-5)
    assert int((i[:, :kk] != i0[:, :kk]).sum()) <= swaps + swaps0 + 2
    print(f"screened {nq}x{ndb}x{dim} {metric}: float64 near-tie swaps screened {swaps} / unscreened {swaps0}")


def test_screened_search_ties_and_duplicates():
    """Exact duplicates of a query's best rows (a tie in every arithmetic): lower index first, as the exact search; 40 copies fit the
    candidate list.  700 copies do not: the call falls back to the unscreened search and gives ITS bits."""
    from anyloc_amd import ops
    qu, db = _data(300, 12000, 2048, 5, planted=False)
    db[100] = 3.0 * qu[0]
    copies = torch.arange(200, 240, device=DEV)
    db[copies] = db[100].clone().expand(len(copies), -1)
    with ops.options(topk_screen=1, topk_h3=1):
        d, i = ops.topk(qu, db, 20, "ip", normalize_db=True)
    assert i[0, 0] == 100 and torch.equal(i[0, 1:20], copies[:19])
    _check(d, i, qu, db, 20, "ip", "duplicates")
    db[3000:3700] = db[100].clone().expand(700, -1)
    with ops.options(topk_screen=1, topk_h3=1):
        ops.profile_enable(True); ops.profile_reset()
        d, i = ops.topk(qu, db, 20, "ip", normalize_db=True)
        torch.cuda.synchronize()
        prof = ops.profile_dump()
        ops.profile_enable(False)
    assert "topk_screen_gemm" in prof and "topk_scores_gemm" in prof            # screened first, then the fallback
    with ops.options(topk_screen=0, topk_h3=1):
        d0, i0 = ops.topk(qu, db, 20, "ip", normalize_db=True)
    assert torch.equal(d, d0) and torch.equal(i, i0)
    _check(d, i, qu, db, 20, "ip", "fallback")


def test_screened_search_through_the_prepared_index_with_its_rows():
    """retrieval.FlatIndex keeps the fp32 rows next to the planes (keep_fp32, the default): its searches run screened
    (anyloc_topk_search_index_rows) and give the lists of the one-shot screened call bit for bit; an index without its rows
    searches unscreened."""
    from anyloc_amd import ops, retrieval
    qu, db = _data(520, 17000, 4096, 9)
    with ops.options(topk_screen=1):
        index = retrieval.FlatIndex(db, "cosine", True, planes=True)
        d, i = index.search(qu, 20)
        d1, i1 = retrieval.search(db, qu, 20)                    # (both normalise the queries the same way first)
        assert torch.equal(d, d1) and torch.equal(i, i1)
        bare = retrieval.FlatIndex(db, "cosine", True, planes=True, keep_fp32=False)
        d2, i2 = bare.search(qu, 20)
    with ops.options(topk_screen=0):
        d3, i3 = retrieval.search(db, qu, 20)
    assert torch.equal(d2, d3) and torch.equal(i2, i3)
    _check(d, i, qu, db, 20, "ip", "indexed")


@pytest.mark.parametrize("nq,ndb,dim,k,metric,qscale", [(300, 20000, 4112, 20, "ip", 1.0),       # odd number of k-blocks: a half-empty last pair
                                                         (300, 20000, 24592, 10, "l2", 1.0),     # two K chunks, the second one k-block long
                                                         (260, 18000, 8192, 20, "l2", 7.5),      # queries that are NOT unit vectors (L2 against normalised rows)
                                                         (260, 18000, 8192, 128, "ip", 0.01)])   # the largest k the screened search serves
def test_screened_search_chunk_edges_and_query_norms(nq, ndb, dim, k, metric, qscale):
    from anyloc_amd import ops
    qu, db = _data(nq, ndb, dim, nq + dim)
    qu = qu * qscale
    with ops.options(topk_screen=1, topk_h3=1):
        ops.profile_enable(True); ops.profile_reset()
        d, i = ops.topk(qu, db, k, metric, normalize_db=True)
        torch.cuda.synchronize()
        prof = ops.profile_dump()
        ops.profile_enable(False)
    assert "topk_screen_gemm" in prof and "topk_scores_gemm" not in prof, sorted(prof)
    q64, d64 = qu.double(), torch.nn.functional.normalize(db.double())
    s = q64 @ d64.t()
    if metric == "l2":
        s = -((q64 * q64).sum(1, keepdim=True) + 1.0 - 2.0 * s)
    o = torch.sort(s, dim=1, descending=True, stable=True)
    got = torch.gather(s, 1, i)
    scale = max(1.0, qscale * qscale) if metric == "l2" else max(qscale, 1e-30)
    tol = (1e-5 if metric == "l2" else 3e-6) * scale
    val = -d.double() if metric == "l2" else d.double()
    assert float((val - got).abs().max()) <= tol, float((val - got).abs().max())
    mism = i != o.indices[:, :k]
    if bool(mism.any()):
        assert float((got[mism] - o.values[:, :k][mism]).abs().max()) <= tol


def test_screened_search_degenerate_rows():
    """Zero rows, rows of tiny and huge norm, one dominant element per row (the leading plane carries almost nothing of the
    rest), a zero query: the bound is measured per row, so none of them may cost a list entry."""
    from anyloc_amd import ops
    qu, db = _data(300, 17000, 4096, 77)
    db[5] = 0.0
    db[6] *= 1e-18
    db[7] *= 1e18
    db[100:400, 17] = 1e4                                  # one element 4 decades above the rest of its row
    qu[3] = 0.0
    qu[4, 33] = 50.0
    qu[4] = torch.nn.functional.normalize(qu[4], dim=0)      # (unit like the others: the bars below are absolute)
    with ops.options(topk_screen=1, topk_h3=1):
        d, i = ops.topk(qu, db, 20, "ip", normalize_db=True)
    assert torch.isfinite(d).all()
    _check(d, i, qu, db, 20, "ip", "degenerate rows")
    with ops.options(topk_screen=0, topk_h3=1):
        d0, i0 = ops.topk(qu, db, 20, "ip", normalize_db=True)
    assert float((d - d0).abs().max()) <= 3e-6


def test_screened_search_more_queries_than_one_operand_image_holds():
    """11 000 queries of 49 152 columns exceed the 2 GiB addressing range of one query image (10 752 rows): two query chunks."""
    from anyloc_amd import ops
    qu, db = _data(11000, 16500, 49152, 3)
    with ops.options(topk_screen=1):
        d, i = ops.topk(qu, db, 20, "ip", normalize_db=True)
    sel = torch.cat([torch.arange(0, 11000, 400, device=DEV), torch.tensor([10751, 10752, 10999], device=DEV)])
    s = qu[sel].double() @ torch.nn.functional.normalize(db.double()).t()
    o = torch.sort(s, dim=1, descending=True, stable=True)
    got = torch.gather(s, 1, i[sel])
    assert float((d[sel].double() - got).abs().max()) <= 3e-6
    mism = i[sel] != o.indices[:, :20]
    if bool(mism.any()):
        assert float((got[mism] - o.values[:, :20][mism]).abs().max()) <= 3e-6


@pytest.mark.parametrize("metric", ["ip", "l2"])
def test_screened_search_on_raw_rows(metric):
    """Without ANYLOC_TOPK_NORMALIZE_DB the rows count with their raw norms (here 0.3 ... 2.3, planted neighbours up to 5.5): the
    bound of a query scales with the LARGEST raw norm of the database; the lists are those of the float64 search over the raw rows."""
    from anyloc_amd import ops
    qu, db = _data(400, 20000, 4096, 21)
    with ops.options(topk_screen=1, topk_h3=1):
        ops.profile_enable(True); ops.profile_reset()
        d, i = ops.topk(qu, db, 20, metric, normalize_db=False)
        torch.cuda.synchronize()
        prof = ops.profile_dump()
        ops.profile_enable(False)
    assert "topk_screen_gemm" in prof and "topk_scores_gemm" not in prof, sorted(prof)
    q64, d64 = qu.double(), db.double()
    s = q64 @ d64.t()
    if metric == "l2":
        s = -((q64 * q64).sum(1, keepdim=True) + (d64 * d64).sum(1)[None, :] - 2.0 * s)
    o = torch.sort(s, dim=1, descending=True, stable=True)
    got = torch.gather(s, 1, i)
    val = -d.double() if metric == "l2" else d.double()
    scale = float(d64.norm(dim=1).max()) ** (2 if metric == "l2" else 1)
    tol = 3e-6 * scale
    assert float((val - got).abs().max()) <= tol, (float((val - got).abs().max()), tol)
    mism = i != o.indices[:, :20]
    if bool(mism.any()):
        assert float((got[mism] - o.values[:, :20][mism]).abs().max()) <= tol
    with ops.options(topk_screen=0, topk_h3=1):
        d0, i0 = ops.topk(qu, db, 20, metric, normalize_db=False)
    assert float((d - d0).abs().max()) <= 2 * tol and int((i != i0).sum()) <= int(mism.sum()) + 4


def test_screened_search_is_reproducible_under_load():
    """The one-shot quantiser adds its residual sums with atomics (their rounding varies run to run, which can move the candidate
    margin by an ulp) and the candidate lists are filled in arrival order: neither may reach the result -- ten searches with
    another stream keeping the chip busy give one set of bits."""
    from anyloc_amd import ops
    qu, db = _data(700, 33000, 4096, 31)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device=DEV)
    with ops.options(topk_screen=1, topk_h3=1):
        d0, i0 = ops.topk(qu, db, 20, "ip", normalize_db=True)
        for rep in range(10):
            with torch.cuda.stream(side):
                for _ in range(6):
                    a @ a
            d, i = ops.topk(qu, db, 20, "ip", normalize_db=True)
            assert torch.equal(i, i0) and torch.equal(d, d0), rep
    torch.cuda.synchronize()
    _check(d0, i0, qu, db, 20, "ip", "under load")
