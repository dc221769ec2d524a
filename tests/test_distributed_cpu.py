"""The N > 1 paths on CPU with gloo, world_size 2: database-sharded retrieval
(all-gather of query descriptors, per-shard top-k, gather + host merge) and row-sharded
k-means (all-reduce of sums / counts).  The device compute step is replaced by the CPU
oracle through the injection points, so what is under test is the collective + merge logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, out_dir)
    finally:
        dist.destroy_process_group()


def _run(fn, tmp_path, world=2):
    mp.spawn(_worker, args=(world, _free_port(), fn, str(tmp_path)), nprocs=world, join=True)


def _search_job(rank, world, out_dir):
    from anyloc_amd import retrieval
    from oracle import faiss_flat
    g = torch.Generator().manual_seed(0)
    db = torch.nn.functional.normalize(torch.randn(101, 24, generator=g))
    db[70] = db[3]                                   # tie across the two shards
    qu = torch.nn.functional.normalize(torch.randn(11, 24, generator=g))
    bounds = [0, 45, 101]
    q_bounds = [0, 4, 11]                            # uneven query split -> padded all-gather
    shard = db[bounds[rank]:bounds[rank + 1]]
    q_loc = qu[q_bounds[rank]:q_bounds[rank + 1]]

    def search_fn(db_s, q_all, k, method, norm, base):
        d, i = faiss_flat.flat_search(q_all, db_s, k, "ip" if method == "cosine" else "l2")
        return d, torch.where(i >= 0, i + base, i)

    for method in ("cosine", "l2"):
        d, i = retrieval.sharded_search(shard, bounds[rank], q_loc, 10, method=method, search_fn=search_fn)
        if rank == 0:
            d_ref, i_ref = faiss_flat.flat_search(qu, db, 10, "ip" if method == "cosine" else "l2")
            assert np.array_equal(i, i_ref.numpy()), method
            np.testing.assert_allclose(d, d_ref.numpy(), atol=1e-5)
        else:
            assert d is None and i is None
    # static shares (what bench.py passes) with the instrumented legs: the same lists, every leg timed on every rank
    legs = {}
    d2, i2 = retrieval.sharded_search(shard, bounds[rank], q_loc, 10, search_fn=search_fn, counts=[4, 7], timings=legs)
    assert {"all_gather_ms", "search_ms", "gather_ms"} <= set(legs) and all(v >= 0 for v in legs.values())
    if rank == 0:
        assert np.array_equal(i2, faiss_flat.flat_search(qu, db, 10, "ip")[1].numpy()) and "merge_ms" in legs
    # the overlapped step (round 6): equal shares, the rank's own queries searched while the others' rows are in flight, the
    # rest afterwards -- the lists of the plain step, in the global query order
    qu2 = torch.nn.functional.normalize(torch.randn(12, 24, generator=g))
    calls = []

    def counting_search(db_s, q_all, k, method, norm, base):
        calls.append(int(q_all.shape[0]))
        return search_fn(db_s, q_all, k, method, norm, base)
    for method in ("cosine", "l2"):
        del calls[:]
        d3, i3 = retrieval.sharded_search(shard, bounds[rank], qu2[6 * rank:6 * rank + 6], 10, method=method,
                                          search_fn=counting_search, counts=[6, 6], overlap=True)
        assert calls == [6, 6], calls                  # own block first, then the other rank's
        if rank == 0:
            d_ref, i_ref = faiss_flat.flat_search(qu2, db, 10, "ip" if method == "cosine" else "l2")
            assert np.array_equal(i3, i_ref.numpy()), method
            np.testing.assert_allclose(d3, d_ref.numpy(), atol=1e-5)
    # (uneven shares fall back to the plain step whatever `overlap` says)
    del calls[:]
    d4, i4 = retrieval.sharded_search(shard, bounds[rank], q_loc, 10, search_fn=counting_search, counts=[4, 7], overlap=True)
    assert calls == [11], calls
    if rank == 0:
        assert np.array_equal(i4, faiss_flat.flat_search(qu, db, 10, "ip")[1].numpy())
    # shares that do not describe this rank's rows are refused BEFORE any collective (no assert: survives python -O)
    with pytest.raises(ValueError):
        retrieval.sharded_search(shard, bounds[rank], q_loc, 10, search_fn=search_fn, counts=[5, 6])
    with pytest.raises(ValueError):
        retrieval.sharded_search(shard, bounds[rank], q_loc, 10, search_fn=search_fn, counts=[11])
    if rank == 0:
        open(os.path.join(out_dir, "search_ok"), "w").write("1")


def _kmeans_job(rank, world, out_dir):
    from anyloc_amd import kmeans as hk, synth
    from oracle import fpk_kmeans

    def step(x, c, mode, want_labels):
        lab = fpk_kmeans.KMeans.cos_sim(x, c).max(dim=-1)[1]
        onehot = (lab[None, :] == torch.arange(c.shape[0])[:, None]).to(x.dtype)
        return onehot @ x, onehot.sum(-1), lab

    x = synth.clustered_tokens(1, 4000, 16, n_modes=5, seed=2, noise=0.5)[0]
    init = x[torch.arange(7) * 500].clone()
    half = 1700                                       # uneven row shards
    x_loc = x[:half] if rank == 0 else x[half:]
    km = hk.KMeans(7, mode="cosine", process_group=dist.group.WORLD, step_fn=step)
    lab = km.fit_predict(x_loc, centroids=init)
    ref = fpk_kmeans.KMeans(7, mode="cosine")
    lab_ref = ref.fit_predict(x, centroids=init.clone())
    assert km.n_iter_ == ref.n_iter_
    assert float((km.centroids - ref.centroids).abs().max()) < 1e-5   # same result as the flat fit
    assert torch.equal(lab, lab_ref[:half] if rank == 0 else lab_ref[half:])
    # no explicit init: rank 0 draws np.random.choice over ALL rows, the draw is broadcast and the rows are collected --
    # the same centroids as the flat fit started from the same NumPy RNG state
    np.random.seed(123)
    km2 = hk.KMeans(7, mode="cosine", process_group=dist.group.WORLD, step_fn=step)
    km2.fit(x_loc)
    np.random.seed(123)
    ref2 = fpk_kmeans.KMeans(7, mode="cosine")
    ref2.fit(x)
    assert km2.n_iter_ == ref2.n_iter_
    assert float((km2.centroids - ref2.centroids).abs().max()) < 1e-5
    if rank == 0:
        open(os.path.join(out_dir, "kmeans_ok"), "w").write("1")


def test_sharded_search_equals_flat_index(tmp_path):
    _run(_search_job, tmp_path)
    assert (tmp_path / "search_ok").exists()


def test_sharded_kmeans_equals_flat_fit(tmp_path):
    _run(_kmeans_job, tmp_path)
    assert (tmp_path / "kmeans_ok").exists()
