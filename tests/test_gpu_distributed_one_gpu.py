"""The N > 1 paths with the REAL HIP kernels, on the one GPU a test box has: two ranks (gloo, world size 2) share
cuda:0 and run the database-sharded retrieval of SURVEY 8(e) / BASELINE configs[2] (query all-gather, per-shard
anyloc_topk with global indices, gather + host merge) and the row-sharded k-means (anyloc_kmeans_step per shard,
all-reduce of sums / counts, broadcast initial draw) -- compared with the single-process flat result.  What RCCL adds
on a multi-GPU node is only the transport of the same collectives (tests/test_distributed_cpu.py covers the host logic
with an injected CPU step)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fn(rank, world, out_dir)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def _run(fn, tmp_path, world=2):
    mp.spawn(_worker, args=(world, _free_port(), fn, str(tmp_path)), nprocs=world, join=True)


def _search_job(rank, world, out_dir):
    from anyloc_amd import retrieval
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(0)
    dim = 32 * 128                                    # few-query path of anyloc_topk (<= 64 queries, dim >= 4096)
    db = torch.randn(3001, dim, generator=g) * (0.5 + torch.rand(3001, 1, generator=g))
    db[2000] = 1.7 * db[3]                            # the same direction in both shards: a tie after normalising
    qu = torch.randn(23, dim, generator=g)
    qu[5] = db[3]
    bounds, q_bounds = [0, 1300, 3001], [0, 9, 23]    # uneven shards, uneven query split (padded all-gather)
    shard = db[bounds[rank]:bounds[rank + 1]].to(dev)
    q_loc = qu[q_bounds[rank]:q_bounds[rank + 1]].to(dev)
    for method in ("cosine", "l2"):
        d, i = retrieval.sharded_search(shard, bounds[rank], q_loc, 10, method=method)
        if rank == 0:
            d_ref, i_ref = retrieval.search(db.to(dev), qu.to(dev), 10, method)      # one flat index, same kernels
            d_ref, i_ref = d_ref.cpu().numpy(), i_ref.cpu().numpy()
            assert np.array_equal(i, i_ref), method
            np.testing.assert_allclose(d, d_ref, atol=2e-6)
            assert {int(i[5, 0]), int(i[5, 1])} == {3, 2000}
            assert i.max() < 3001 and i.min() >= 0
        else:
            assert d is None and i is None
    # many queries (panel-GEMM path of anyloc_topk) against the same shards
    qu2 = torch.randn(150, dim, generator=g)
    d, i = retrieval.sharded_search(shard, bounds[rank], qu2[75 * rank:75 * (rank + 1)].to(dev), 5)
    if rank == 0:
        d_ref, i_ref = retrieval.search(db.to(dev), qu2.to(dev), 5)
        assert np.array_equal(i, i_ref.cpu().numpy())
        open(os.path.join(out_dir, "search_ok"), "w").write("1")


def _kmeans_job(rank, world, out_dir):
    import utilities
    from anyloc_amd import kmeans as hk, synth
    dev = torch.device("cuda", 0)
    x = torch.nn.functional.normalize(synth.clustered_tokens(1, 20000, 384, n_modes=12, seed=2, noise=0.6)[0])
    half = 8500                                       # uneven row shards
    x_loc = (x[:half] if rank == 0 else x[half:]).to(dev)
    np.random.seed(7)
    km = hk.KMeans(12, mode="cosine", process_group=dist.group.WORLD)        # init drawn by rank 0, broadcast
    lab = km.fit_predict(x_loc)
    np.random.seed(7)
    ref = hk.KMeans(12, mode="cosine")                                        # the flat fit, same kernels, one process
    lab_ref = ref.fit_predict(x.to(dev))
    assert km.n_iter_ == ref.n_iter_
    assert float((km.centroids.cpu() - ref.centroids.cpu()).abs().max()) < 1e-5
    mine = lab_ref[:half] if rank == 0 else lab_ref[half:]
    assert int((lab.cpu() != mine.cpu()).sum()) <= 2
    # through the class surface: VLAD.fit(shard, process_group=...) == VLAD.fit(all rows)
    np.random.seed(11)
    v = utilities.VLAD(12, 384, cache_dir=None)
    v.fit(x_loc, process_group=dist.group.WORLD)
    np.random.seed(11)
    v0 = utilities.VLAD(12, 384, cache_dir=None)
    v0.fit(x.to(dev))
    assert float((v.c_centers.cpu() - v0.c_centers.cpu()).abs().max()) < 1e-5
    if rank == 0:
        open(os.path.join(out_dir, "kmeans_ok"), "w").write("1")


def test_sharded_search_two_ranks_one_gpu(tmp_path):
    _run(_search_job, tmp_path)
    assert (tmp_path / "search_ok").exists()


def test_sharded_kmeans_two_ranks_one_gpu(tmp_path):
    _run(_kmeans_job, tmp_path)
    assert (tmp_path / "kmeans_ok").exists()
