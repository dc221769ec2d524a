"""Round-4 GPU tests: retrieval index identity at the full BASELINE.json configs[1] size and on one configs[2] panel
against an exact float64 search, the host <-> device transfers, the reference scripts' CPU-tensor call pattern through
``VLAD.generate_multi`` / ``get_top_k_recall``, and an 8-rank run of ``bench.py --workload config3`` on one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def _vlad_like(n, k, d, seed):
    """Rows shaped like VLADs: per-cluster unit blocks, globally normalised (bench.py's synthetic database recipe)."""
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    out = torch.empty(n, k * d, dtype=torch.float32, device=DEV)
    for s in range(0, n, 1000):
        e = min(n, s + 1000)
        blk = torch.nn.functional.normalize(torch.randn(e - s, k, d, generator=g, device=DEV), dim=-1) / (k ** 0.5)
        out[s:e] = blk.reshape(e - s, k * d)
    return out


def _float64_flat(qu, db, k, chunk=128):
    """Exact flat inner-product search of the L2-normalised operands in float64 on the device (the checker): indices with
    faiss' tie rule (lower index first) and the float64 score matrix chunks for tie-aware comparison."""
    dbn = torch.nn.functional.normalize(db.double(), dim=1)
    for s0 in range(0, qu.shape[0], chunk):
        sc = torch.nn.functional.normalize(qu[s0:s0 + chunk].double(), dim=1) @ dbn.T
        yield s0, sc, torch.sort(-sc, dim=1, stable=True)[1][:, :k]


def _assert_identity(qu, db, dist, idx, k, tol=3e-6):
    mism = swaps = 0
    for s0, sc, order in _float64_flat(qu, db, k):
        ours = idx[s0:s0 + order.shape[0]]
        got, want = torch.gather(sc, 1, ours), torch.gather(sc, 1, order)
        diff = ours != order
        near = (got - want).abs() <= tol
        swaps += int((diff & near).sum())
        mism += int((diff & ~near).sum())
        assert float((dist[s0:s0 + order.shape[0]].double() - got).abs().max()) <= tol
        # a swap only ever exchanges near-tied neighbours: as SETS the lists differ at most at the k-th rank
    assert mism == 0, f"{mism} indices differ from the float64 search outside {tol} ties ({swaps} near-tie swaps)"
    return swaps


def test_topk_index_identity_at_the_full_config2_size():
    """1 000 queries x 10 000 rows x 49 152 columns (BASELINE.json configs[1], the reference's utilities.py:433-450 at its
    headline size): every index of the top-20 equals the exact float64 flat search (ties -> lower index), distances within
    3e-6, and a 48-query slice equals the faiss restatement of the oracle on the CPU."""
    from anyloc_amd import retrieval
    from oracle import faiss_flat
    db = _vlad_like(10000, 32, 1536, 1)
    qu = _vlad_like(1000, 32, 1536, 2)
    src = torch.arange(1000, device=DEV) * 7 + 3
    qu[:600] = 0.6 * db[src[:600]] + 0.4 * qu[:600]              # neighbours at every similarity level
    qu[600:700] = db[src[600:700]]                               # exact copies (cosine 1)
    qu[700:720] = 0.5 * (db[src[700:720]] + db[src[700:720] + 1])  # two near-equal neighbours
    db[9000:9010] = db[8000:8010]                                # duplicated rows: exact ties, the lower index first
    for lo, hi in ((0, 1000), (0, 61), (100, 356)):             # many-query panels, the few-query kernel, a mid-size call
        d, i = retrieval.search(db, qu[lo:hi], 20)
        swaps = _assert_identity(qu[lo:hi], db, d, i, 20)
        assert swaps <= (hi - lo) // 10
    d, i = retrieval.search(db, qu[640:688], 20)
    dr, ir = faiss_flat.flat_search(torch.nn.functional.normalize(qu[640:688].cpu()), torch.nn.functional.normalize(db.cpu()), 20)
    assert torch.equal(i[:, 0].cpu(), ir[:, 0])
    assert float((d.cpu() - dr).abs().max()) <= 3e-6
    both = (i.cpu() == ir)
    assert float(both.float().mean()) > 0.98                     # fp32 CPU search vs fp32-accurate GPU search: near-ties may swap


def test_topk_index_identity_on_a_config3_panel():
    """2 048 queries against one 8 192-row panel of 49 152 columns (the unit of work of the many-query path at configs[2]:
    two-term fp16 score panels on the 16x16x32 MFMA kernel, K-chunked accumulation): identical to the float64 search."""
    from anyloc_amd import retrieval
    db = _vlad_like(8192, 32, 1536, 3)
    qu = _vlad_like(2048, 32, 1536, 4)
    src = torch.arange(1024, device=DEV) * 5 + 1
    qu[:1024] = 0.7 * db[src] + 0.3 * qu[:1024]
    d, i = retrieval.search(db, qu, 20)
    assert torch.equal(i[:1024, 0], src)
    _assert_identity(qu, db, d, i, 20)


def test_host_device_transfers_round_trip():
    """ops.to_device / ops.to_host (the product's only host <-> device copies) move bytes unchanged: small and large tensors,
    non-contiguous sources, other dtypes, pinned sources."""
    from anyloc_amd import ops as staging
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator().manual_seed(0)
    for shape, dtype in (((7, 13), torch.float32), ((3, 322, 322), torch.float32), ((40, 529, 1536), torch.float32),
                         ((9_000_001,), torch.int64), ((5, 3), torch.float64), ((0, 4), torch.float32)):
        t = (torch.randn(shape, generator=g) * 100).to(dtype)
        on = staging.to_device(t, dev)
        assert on.is_cuda and on.dtype == dtype and torch.equal(on.cpu(), t)
        back = staging.to_host(on)
        assert back.device.type == "cpu" and torch.equal(back, t)
    t = torch.randn(300, 400, generator=g)
    nc = t.t()[5:200:3]                                          # non-contiguous view
    assert torch.equal(staging.to_device(nc, dev).cpu(), nc)
    pinned = torch.randn(1000, 100, generator=g).pin_memory()
    assert torch.equal(staging.to_device(pinned, dev).cpu(), pinned)


def test_cpu_tensor_call_pattern_equals_the_device_path():
    """What scripts/dino_v2_vlad.py does (:236-260, :372): CPU patch descriptors -> VLAD.generate_multi -> CPU VLADs ->
    get_top_k_recall with a CPU database.  Bitwise the results of the same calls on device tensors."""
    import utilities
    g = torch.Generator().manual_seed(3)
    tok = torch.nn.functional.normalize(torch.randn(37, 529, 1536, generator=g), dim=-1)
    vlad = utilities.VLAD(32, 1536, cache_dir=None)
    np.random.seed(1)
    vlad.fit(tok[:8].reshape(-1, 1536))
    v_cpu = vlad.generate_multi(tok)
    v_dev = vlad.generate_multi(tok.to(DEV))
    assert v_cpu.device.type == "cpu" and v_dev.is_cuda and torch.equal(v_cpu, v_dev.cpu())
    vlad.HOST_CHUNK_BYTES = 10 * 529 * 1536 * 4                  # the chunked host path (pieces of 10 images): every piece runs with
    v_chunked = vlad.generate_multi(tok)                         # the whole batch's workgroups-per-image count -> the SAME bits
    assert torch.equal(v_chunked, v_cpu)
    db = _vlad_like(3000, 32, 1536, 9).cpu()
    db[5:42] = v_cpu
    gt = np.empty(37, dtype=object)
    for j in range(37):
        gt[j] = np.array([5 + j])
    d1, i1, r1 = utilities.get_top_k_recall([1, 5], db, v_cpu, gt)
    d2, i2, r2 = utilities.get_top_k_recall([1, 5], db.to(DEV), v_dev, gt)
    assert i1.device.type == "cpu" and torch.equal(i1, i2.cpu()) and torch.equal(d1, d2.cpu())
    assert r1 == r2 and r1[1] == 1.0


def test_bench_config3_eight_ranks_on_one_gpu():
    """``bench.py --workload config3 --gpus 8`` at a reduced shard (8 x 2 048 rows, 512 queries) with the ranks sharing the one
    GPU over gloo: the query all-gather with static counts, eight per-shard searches with global indices, the single packed
    gather and the host merge -- the planted neighbour of every rank-0 query is found in ITS shard."""
    env = dict(os.environ, ANYLOC_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "config3", "--gpus", "8", "--steps", "1",
                          "--warmup", "0", "--queries", "512", "--shard-rows", "2048"], env=env, capture_output=True, text=True,
                         timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["unit"] == "queries/s" and out["value"] > 0
    assert out["planted_neighbours_found"] is True
    assert out["config"]["db_rows_total"] == 8 * 2048 and out["config"]["queries"] == 512


# ---------------------------------------------------------------------------------- few-query scores, running row scale
def _check_vs_float64(d, i, qu, db, k, metric, norm, tol=3e-6):
    """(dist, idx) of ops.topk against the exact float64 search on the device: tie-aware index identity, distance error."""
    q64, d64 = qu.double(), db.double()
    if norm:
        q64, d64 = torch.nn.functional.normalize(q64, dim=1), torch.nn.functional.normalize(d64, dim=1)
    if metric == "ip":
        sc = q64 @ d64.T
        order = torch.sort(-sc, dim=1, stable=True)[1][:, :k]
    else:
        sc = (q64 * q64).sum(1)[:, None] + (d64 * d64).sum(1)[None] - 2 * q64 @ d64.T
        order = torch.sort(sc, dim=1, stable=True)[1][:, :k]
    got, want = torch.gather(sc, 1, i), torch.gather(sc, 1, order)
    scale = float(sc.abs().max()) if not norm else 1.0
    assert int(((i != order) & ((got - want).abs() > tol * scale)).sum()) == 0
    assert float((d.double() - got).abs().max()) <= tol * scale, float((d.double() - got).abs().max())


@pytest.mark.parametrize("metric,norm", [("ip", True), ("ip", False), ("l2", True), ("l2", False)])
def test_topk_few_queries_two_fp16_planes_running_scale(metric, norm):
    """Option topk_fewq_x6 = 2 (csrc/scores_h3.hip): <= 64 queries, database rows split on the fly into two fp16 planes under a
    power-of-two scale that RUNS along each row -- ragged row tail, 61 and 3 queries, row norms from the same pass; rows
    whose magnitude grows along K (a new maximum, i.e. a rescale of the accumulators, in most slabs), rows with one late
    spike, all-zero and tiny rows.  Against float64, and reproducible run to run."""
    from anyloc_amd import ops
    g = torch.Generator(device=DEV)
    g.manual_seed(21)
    dim, ndb = 8192, 10037
    scale = 0.05 + torch.rand(ndb, 1, generator=g, device=DEV) * 20.0
    db = torch.randn(ndb, dim, generator=g, device=DEV) * scale
    ramp = torch.exp2(torch.arange(dim, device=DEV) / dim * 12.0)              # magnitudes grow 4096 x along the row
    db[100:164] *= ramp
    db[200:232, 7000] = 5e3                                                     # one late spike
    db[300:310] = 0.0
    db[320:330] *= 1e-30
    db[5000] = db[12]
    for nq, k in ((61, 20), (3, 7)):
        qu = torch.randn(nq, dim, generator=g, device=DEV)
        pick = torch.arange(min(nq, 30), device=DEV) * 301 + 12
        qu[: len(pick)] = db[pick] + 0.3 * scale[pick] * torch.randn(len(pick), dim, generator=g, device=DEV)
        qu[-1] = db[130]                                                        # a ramp row as the query
        qd = torch.nn.functional.normalize(qu) if norm else qu
        with ops.options(topk_fewq_x6=2):
            d, i = ops.topk(qd, db, k, metric, normalize_db=norm)
            d1, i1 = ops.topk(qd, db, k, metric, normalize_db=norm)
        assert torch.equal(i, i1) and torch.equal(d, d1)
        with ops.options(topk_fewq_x6=2, topk_fewq_qdma=0):                     # queries split per slab by the staging lanes: the same bits
            d2, i2 = ops.topk(qd, db, k, metric, normalize_db=norm)
        assert torch.equal(i, i2) and torch.equal(d, d2)
        _check_vs_float64(d, i, qu, db, k, metric, norm)
        if norm or metric == "l2":                                              # (the raw inner product prefers the huge ramp rows)
            assert int(i[0, 0]) == 12 and int(i[0, 1]) == 5000                  # the duplicated row: lower index first
        with ops.options(topk_fewq_x6=1):
            d0, i0 = ops.topk(qd, db, k, metric, normalize_db=norm)
        assert float((i0 != i).float().mean()) < 0.003


def test_topk_few_queries_running_scale_self_match_long_rows():
    """131 072 columns (the ViT-L two-tap VLAD), queries that ARE database rows, all-positive rows: every product is a
    square, whatever is dropped adds up.  Cosine 1 to 2e-6 and the exact top-5."""
    from anyloc_amd import ops
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    dim, ndb, nq = 131072, 260, 8
    db = torch.nn.functional.normalize(torch.randn(ndb, dim, generator=g, device=DEV).abs() + 0.5, dim=1) * 2.5
    qu = db[:nq].clone()
    with ops.options(topk_fewq_x6=2):
        d, i = ops.topk(torch.nn.functional.normalize(qu), db, 5, "ip", normalize_db=True)
    assert torch.equal(i[:, 0].cpu(), torch.arange(nq))
    _check_vs_float64(d, i, qu, db, 5, "ip", True, tol=2e-6)


def test_reference_script_call_pattern_stage_on_a_small_model():
    """bench.py's `script_path_vitg` stage -- per image ``ext(img[None].to(device)).cpu()``, ``VLAD.generate_multi`` on the CPU
    tensor, ``get_top_k_recall`` on CPU tensors (scripts/dino_v2_vlad.py:164-188, :236-260, :372) -- on ViT-S/14 at 224 x 224:
    the stage's own check against the batched device path must hold (tokens, cluster ids, VLADs, top-1)."""
    import bench
    import utilities
    from anyloc_amd import synth, weights
    name = "dinov2_vits14"
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 2, device=DEV, depth=10))
    try:
        ext = utilities.DinoV2ExtractFeatures(name, 9, "value", device=DEV)
        db_img, qu_img, gt = synth.synthetic_places(24, 24, 224, 224, seed=3, device=DEV)
        vlad = utilities.VLAD(8, None, cache_dir=None)
        tok = ext(db_img)
        np.random.seed(3)
        vlad.fit(tok.reshape(-1, tok.shape[-1]))
        db = torch.nn.functional.normalize(torch.randn(400, 8 * 384, generator=torch.Generator().manual_seed(1)), dim=1).to(DEV)
        db[:24] = vlad.generate_multi(tok)
        res = bench.stage_script_path(ext, vlad, db, qu_img, gt, n_img=24)
        assert res["oracle_ok"], res
        assert res["vs_batched_device_path"]["cpu_tensor_path_bitwise_equals_device_call"]
        assert res["images_per_s"] > 0
    finally:
        weights.unregister_state_dict(name)


def test_reference_call_pattern_driver_through_anyloc_amd_run(tmp_path):
    """``python -m anyloc_amd.run tests/drivers/vlad_driver_standin.py`` in a fresh interpreter on the GPU box: the reference
    driver's call sequence on the ``utilities`` surface (scripts/dino_v2_vlad.py:157-188, :195-260, :372-376) -- tyro CLI and
    torchvision transforms through the shims, one image per extractor call with ``.to(device)`` / ``.cpu()``, ``VLAD.fit`` +
    ``generate_multi`` with cache ids (``c_centers.pt``, ``_r/_l`` files), ``get_top_k_recall`` on CPU tensors -- twice: the
    second run restores the vocabulary and the VLADs from the cache and must print the same recalls and top-1 indices.
    (The reference's own script runs the same way wherever its tree exists: tests/test_gpu_round3.py.)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synth_dataset
    make_synth_dataset.write(str(tmp_path / "data"), "st_lucia", n_db=6, n_qu=4, h=112, w=140)
    env = dict(os.environ, ANYLOC_SYNTHETIC_WEIGHTS="0", PYTHONPATH=ROOT)
    outs = []
    for _ in range(2):
        res = subprocess.run([sys.executable, "-m", "anyloc_amd.run", os.path.join(ROOT, "tests", "drivers", "vlad_driver_standin.py"),
                              "--data-dir", str(tmp_path / "data"), "--cache-dir", str(tmp_path / "cache"), "--num-clusters", "4"],
                             env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0 and "Traceback" not in res.stdout + res.stderr, (res.stdout[-2000:], res.stderr[-2000:])
        outs.append(res.stdout)
    for out in outs:
        assert "Database VLADs shape: torch.Size([6, 1536])" in out and "Query VLADs shape: torch.Size([4, 1536])" in out
        assert "device of the results: cpu cpu" in out
    pick = lambda out: [l for l in out.splitlines() if l.startswith("R@") or l.startswith("top-1")]
    assert pick(outs[0]) == pick(outs[1]) and len(pick(outs[0])) == 4
    assert "Using cached cluster centers" in outs[1] and "Using cached cluster centers" not in outs[0]
    rec = [float(l.split(":")[1]) for l in pick(outs[0])[:3]]
    assert 0.5 <= rec[0] <= rec[1] <= rec[2] <= 1.0, outs[0][-800:]    # query q depicts place q (random-weight ViT-S: mostly found)
    pts = [f for dp, _, fs in os.walk(tmp_path / "cache") for f in fs if f.endswith(".pt")]
    assert "c_centers.pt" in pts and any(f.endswith("_r.pt") for f in pts) and any(f.endswith("_l.pt") for f in pts)


@pytest.mark.parametrize("name,batch,hw", [("dinov2_vits14", 1, (224, 322)), ("dinov2_vitg14", 3, (322, 322))])
def test_patch_embedding_on_the_fp16_gemm_agrees_with_the_fp32_gemm(name, batch, hw):
    """fp16 mode runs the patch embedding (conv 14x14 / 14 = a GEMM over 588-element patches, zero-padded to 592) on the
    two-term fp16 GEMM, the gathered patches and the weights quantised like every other operand (22 bits).  Against the fp32
    matrix-core GEMM (option h3_patch = 0) the layer-0 tokens -- the embedding after one block -- agree to 2e-6 of their
    scale, for unnormalised pixel values too (0 .. 255: the row scale is a power of two of the row's own maximum), and the
    result is the same run to run."""
    import utilities
    from anyloc_amd import ops, synth, weights
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 7, device=DEV, depth=1))
    try:
        ext = utilities.DinoV2ExtractFeatures(name, 0, "token", device=DEV)
        ext.dino_model.ffn_check = False
        g = torch.Generator().manual_seed(5)
        for img in (torch.randn(batch, 3, *hw, generator=g), torch.rand(batch, 3, *hw, generator=g) * 255.0):
            img = img.to(DEV)
            with ops.options(h3_patch=0):
                want = ext(img).clone()
            got = ext(img).clone()
            assert torch.isfinite(got).all()
            assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
            assert torch.equal(got, ext(img))
    finally:
        weights.unregister_state_dict(name)


@pytest.mark.parametrize("workload", ["config2", "config3"])
def test_sharded_step_on_real_rccl_with_one_rank(workload):
    """``ANYLOC_DIST_FORCE=1 python bench.py --gpus 1``: the process group is created with backend "nccl" (= RCCL) although
    there is one rank, and the step takes the SHARDED route -- all-gather of the query VLADs, per-shard top-k on the HIP
    kernels, gather of the [Q, k] lists to rank 0, host merge (retrieval.sharded_search) -- so every collective of the
    N > 1 path (device tensors, their dtypes, the barrier + MAX-over-ranks timing) executes on real RCCL, which a one-GPU
    box otherwise never does.  Recalls / planted neighbours as in the single-process run."""
    env = dict(os.environ, ANYLOC_DIST_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ANYLOC_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"]
    if workload == "config2":
        cmd += ["--batch", "8", "--no-cpu-baseline", "--no-modes", "--no-stages"]
    else:
        cmd += ["--workload", "config3", "--queries", "300", "--shard-rows", "4096"]
    def run(e):
        res = subprocess.run(cmd, env=e, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, res.stdout[-2000:]
        return json.loads(lines[0])
    out = run(env)
    assert out["n_gpus"] == 1 and out["value"] > 0
    if workload == "config2":
        assert out["config"]["parallelism"] == "dp1+db-shard1"
        plain = run({k: v for k, v in env.items() if k != "ANYLOC_DIST_FORCE"})  # the single-process route: same recalls
        assert plain["config"]["parallelism"] == "single" and plain["recall"] == out["recall"]
    else:
        assert out["config"]["parallelism"] == "db-shard1" and out["planted_neighbours_found"] is True


_RCCL_ONE_RANK_JOB = r"""
import numpy as np, torch, torch.distributed as dist
from anyloc_amd import kmeans as hk, retrieval, synth
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
db, qu = torch.randn(1501, 4096, generator=g).to(dev), torch.randn(13, 4096, generator=g).to(dev)
d, i = retrieval.sharded_search(db, 0, qu, 7)                       # all-gather of the queries, top-k, gather, host merge
d_ref, i_ref = retrieval.search(db, qu, 7)
assert np.array_equal(i, i_ref.cpu().numpy()) and np.allclose(d, d_ref.cpu().numpy(), atol=1e-6)
x = synth.clustered_tokens(1, 3000, 384, n_modes=6, seed=2, noise=0.5)[0].to(dev)
np.random.seed(11)
km = hk.KMeans(6, mode="cosine", process_group=dist.group.WORLD)    # sharded init (broadcast + all-reduce), all-reduce per iteration
km.fit(x)
np.random.seed(11)
flat = hk.KMeans(6, mode="cosine")
flat.fit(x)
assert km.n_iter_ == flat.n_iter_ and torch.equal(km.centroids, flat.centroids)
assert torch.equal(km.predict(x), flat.predict(x))
dist.barrier()
dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
"""


def test_sharded_search_and_sharded_kmeans_on_real_rccl_with_one_rank():
    """The library-level N > 1 entry points -- ``retrieval.sharded_search`` and ``KMeans(process_group=...)`` -- on a process
    group whose backend IS RCCL (one rank, cuda:0): broadcast, all-reduce, all-gather and gather of device tensors all run,
    and the results are those of the unsharded calls (bitwise for k-means: one shard = the flat fit)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK_JOB], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "RCCL-ONE-RANK-OK" in res.stdout, (res.stdout[-1500:], res.stderr[-3000:])
