"""``get_top_k_recall`` (reference ``utilities.py:390-469``) on the HIP top-k kernel,
plus the database-sharded multi-GPU search of SURVEY 8(e).

The faiss flat index (``IndexFlatIP`` / ``IndexFlatL2`` add + search,
``utilities.py:439-450``) is replaced by csrc/topk.hip; the recall loop
(``:453-468``) is tiny host-side set arithmetic and stays in NumPy as in the
reference.
"""
from typing import List, Tuple

import numpy as np
import torch

from . import _lib, ops


def recalls_from_indices(top_k, indices, gt_pos, use_percentage=True, sub_sample_db=1,
                         sub_sample_qu=1):
    """Recall@k from retrieved indices (reference utilities.py:451-468)."""
    indices = np.asarray(indices)
    recalls = dict(zip(top_k, [0] * len(top_k)))
    for i_qu, qu_retr in enumerate(indices):
        for i_rec in top_k:
            correct_retr = gt_pos[i_qu * sub_sample_qu]
            if np.any(np.isin(qu_retr[:i_rec] * sub_sample_db, correct_retr)):
                recalls[i_rec] += 1
    if use_percentage:
        for k in recalls:
            recalls[k] /= len(indices)
    return recalls


class FlatIndex:
    """The flat index of ``get_top_k_recall`` kept between searches -- faiss' ``index.add(db)`` (reference
    ``utilities.py:441-442, :446-447``) apart from ``index.search(qu, k)`` (``:450``).  ``get_top_k_recall`` builds and
    drops its index per call, as the reference does; a caller that searches ONE database several times (a resident shard
    of the sharded search, the own-queries / other-ranks'-queries halves of an overlapped step) builds a ``FlatIndex``
    once: the rows as the two-plane fp16 operand images of the score GEMM + their scales and sums of squares
    (``anyloc_topk_index_build``; 4 bytes per element, the size of the fp32 rows), which every later search that
    ``anyloc_topk`` would score on its fp16 panels (``anyloc_topk_path``: many queries) reads instead of re-quantising the
    database panel by panel (14 ms of a 320 ms retrieval on a 125 000 x 49 152 shard); the other shapes (few queries
    streaming the rows once, small problems on the fp32-MFMA panels) keep their path.  Results are therefore those of
    ``search`` bit for bit (same kernels, same operands).

    ``planes``: "auto" builds the images when the shape is served (dim % 16 == 0) and the device has the memory to spare,
    True insists, False never (every search quantises on the fly, as ``search``).  ``keep_fp32=False`` drops the reference
    to the fp32 rows once the images exist (every search then runs on the panels)."""

    def __init__(self, db, method="cosine", norm_descs=True, planes="auto", keep_fp32=True):
        if method not in ("cosine", "l2"):
            raise NotImplementedError(f"Method: {method}")
        dev = _lib.require_gpu()
        self.method, self.norm_descs = method, bool(norm_descs)
        self.db = ops._f32c(torch.as_tensor(db), dev)
        self.ntotal, self.dim = int(self.db.shape[0]), int(self.db.shape[1])
        self.planes = None
        nbytes = ops.topk_index_bytes(self.ntotal, self.dim) if self.ntotal else 0
        if planes == "auto":
            planes = nbytes > 0 and torch.cuda.mem_get_info(dev)[0] > nbytes + (8 << 30)
        if planes:
            self.planes = ops.topk_index_build(self.db)
            if not keep_fp32:
                self.db = None

    @property
    def has_planes(self):
        return self.planes is not None

    def search(self, qu, k):
        """(dist, idx) device tensors of the top ``k`` rows for every query (``index.search``)."""
        dev = _lib.require_gpu()
        qu_d = ops._f32c(torch.as_tensor(qu), dev)
        if qu_d.dim() == 1:
            qu_d = qu_d.unsqueeze(0)
        if self.norm_descs:
            qu_d = ops.l2norm_rows(qu_d)
        metric = "ip" if self.method == "cosine" else "l2"
        if self.planes is not None and (self.db is None or
                                        _lib.load().anyloc_topk_path(int(qu_d.shape[0]), self.ntotal, self.dim) == 2):
            return ops.topk_indexed(qu_d, self.planes, self.ntotal, int(k), metric, normalize_db=self.norm_descs, db=self.db)
        return ops.topk(qu_d, self.db, int(k), metric, normalize_db=self.norm_descs)


def search(db, qu, k, method="cosine", norm_descs=True):
    """Normalise (optionally) and search: returns device tensors (dist, idx).  ``db``: the rows, or a ``FlatIndex``."""
    if isinstance(db, FlatIndex):
        if (db.method, db.norm_descs) != (method, bool(norm_descs)):
            raise ValueError(f"FlatIndex built for ({db.method}, norm_descs={db.norm_descs}), searched with ({method}, {norm_descs})")
        return db.search(qu, k)
    dev = _lib.require_gpu()
    db_d, qu_d = ops._f32c(db, dev), ops._f32c(qu, dev)
    if method == "cosine":
        metric = "ip"
    elif method == "l2":
        metric = "l2"
    else:
        raise NotImplementedError(f"Method: {method}")
    if norm_descs:
        # F.normalize of both sides (reference utilities.py:436-437): the queries are normalised here, the database
        # rows inside the search (their norms come out of the scoring pass; no normalised copy of the database)
        qu_d = ops.l2norm_rows(qu_d)
    return ops.topk(qu_d, db_d, int(k), metric, normalize_db=bool(norm_descs))


def get_top_k_recall(top_k: List[int], db: torch.Tensor, qu: torch.Tensor, gt_pos: np.ndarray,
                     method: str = "cosine", norm_descs: bool = True, use_gpu: bool = False,
                     use_percentage: bool = True, sub_sample_db: int = 1,
                     sub_sample_qu: int = 1) -> Tuple[np.ndarray, np.ndarray, dict]:
    """
        Given a database and query (or queries), get the top 'k' retrievals
        (closest in database for each query) as indices (in database),
        distances, and recalls.  Arguments and return values as in the
        reference; ``use_gpu`` is accepted and ignored (the search always runs
        on the GPU here).
    """
    db, qu = torch.as_tensor(db), torch.as_tensor(qu)
    if len(qu.shape) == 1:
        qu = qu.unsqueeze(0)
    if method not in ("cosine", "l2"):
        raise NotImplementedError(f"Method: {method}")
    home = qu.device
    distances, indices = search(db, qu, max(top_k), method, norm_descs)
    distances, indices = ops.to_home(distances, home), ops.to_home(indices, home)
    idx_host = indices if indices.device.type == "cpu" else ops.to_home(indices, "cpu")
    recalls = recalls_from_indices(top_k, idx_host.numpy(), gt_pos, use_percentage,
                                   sub_sample_db, sub_sample_qu)
    return distances, indices, recalls


# ------------------------------------------------------------------ multi-GPU
def merge_shard_topk(dists, idxs, k, metric="ip"):
    """k-way merge of per-shard top-k lists (host).  ``dists``/``idxs``: lists of
    [nq,k] arrays with GLOBAL indices.  Result is identical to one flat index over
    the concatenated database; ties -> lower global index (as the kernel)."""
    d = np.concatenate([np.asarray(x) for x in dists], axis=1)
    i = np.concatenate([np.asarray(x) for x in idxs], axis=1)
    key = -d if metric == "ip" else d
    pad = i < 0
    key = np.where(pad, np.inf, key)
    tie = np.where(pad, np.iinfo(np.int64).max, i)
    order = np.lexsort((tie, key), axis=1)[:, :k]
    return np.take_along_axis(d, order, 1), np.take_along_axis(i, order, 1)


def comm_device(group, like):
    """Device whose tensors the collectives of ``group`` move: the GPU for RCCL (backend "nccl" rejects CPU tensors),
    the host for gloo."""
    import torch.distributed as dist
    return torch.device("cpu") if dist.get_backend(group) == "gloo" else torch.device(like)


def gather_rows(local, counts, rank, group, comm):
    """All-gather of row blocks [counts[r], D] into ONE [sum(counts), D] buffer on ``comm`` without padded or
    per-rank staging copies: equal shares are a single ``all_gather_into_tensor`` straight into the result; uneven
    shares (Q % world != 0) are one broadcast per rank into that rank's row range of the result (same bytes on the wire
    as the all-gather; the only copy is each rank's own share into its slot)."""
    import torch.distributed as dist
    world = len(counts)
    total = sum(counts)
    out = torch.empty(total, local.shape[1], dtype=torch.float32, device=comm)
    src = local.to(comm, torch.float32).contiguous()
    if len(set(counts)) == 1:
        if total:
            dist.all_gather_into_tensor(out, src, group=group)
        return out
    off = 0
    for r in range(world):
        view = out[off:off + counts[r]]
        if r == rank:
            view.copy_(src)
        if counts[r]:
            dist.broadcast(view, src=r if group is None or group is dist.group.WORLD else dist.get_global_rank(group, r),
                           group=group)
        off += counts[r]
    return out


def sharded_search(db_shard, shard_base, qu_local, k, method="cosine", norm_descs=True,
                   group=None, search_fn=None, counts=None, timings=None, overlap="auto"):
    """Database-sharded retrieval, one process per GPU.

    Every rank owns ``db_shard`` (rows ``shard_base ...`` of the global database)
    and a slice ``qu_local`` of the queries.  Step 1: all-gather the query
    descriptors (RCCL over xGMI) so each rank holds all queries; step 2: local
    top-k on the shard with global indices; step 3: gather the [nq,k] lists on
    rank 0 and merge on the host.  Returns (dist, idx) numpy arrays on rank 0,
    (None, None) elsewhere.  ``search_fn`` is injectable for CPU tests of the
    collective / merge logic.  ``counts`` (optional): the query rows of every rank when the caller knows them (static
    shares, as in bench.py) -- saves the per-call all-gather of the counts and its host sync.  ``timings`` (optional dict):
    an INSTRUMENTED call -- the device is drained after every leg and the legs' wall times land in it as
    ``all_gather_ms`` / ``search_ms`` / ``gather_ms`` / ``merge_ms`` (bench.py reports them from an untimed step; the
    timed steps run without the drains).

    ``db_shard`` may be a ``FlatIndex`` (the resident shard prepared once).  ``overlap`` (round 6): the rank's OWN queries
    are searched while the all-gather of the others' is in flight (``async_op``: RCCL runs it on its own stream over xGMI)
    and the rest afterwards -- two searches over the shard instead of one, so it pays only where a second pass costs no
    second quantisation of the database: "auto" = when ``db_shard`` is a ``FlatIndex`` with prepared planes, the shares
    are equal (one ``all_gather_into_tensor``) and the call is not an instrumented one.  Same lists either way."""
    import time

    import torch.distributed as dist

    def lap(name, t0):
        if timings is None:
            return t0
        if qu_local.is_cuda:
            torch.cuda.synchronize(qu_local.device)
        t1 = time.perf_counter()
        timings[name] = timings.get(name, 0.0) + (t1 - t0) * 1e3
        return t1
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    # RCCL moves device tensors over xGMI; a gloo group (CPU tests, single-GPU tests) is staged through the host
    comm = comm_device(group, qu_local.device)
    if counts is None:
        cnt = torch.zeros(world, dtype=torch.int64, device=comm)
        dist.all_gather_into_tensor(cnt, torch.tensor([qu_local.shape[0]], dtype=torch.int64, device=comm), group=group)
        counts = [int(c) for c in cnt.cpu()]
    else:
        counts = [int(c) for c in counts]
        if len(counts) != world or counts[rank] != qu_local.shape[0]:
            # (not an assert: under -O a mismatch would reach the collective and hang or corrupt rows)
            raise ValueError(f"counts must list every rank's query rows: got {counts} for world size {world}, "
                             f"rank {rank} holds {qu_local.shape[0]} rows")
    t0 = lap("setup_ms", time.perf_counter()) if timings is not None else 0.0

    def run_search(q):
        if search_fn is None:
            dd, ii = search(db_shard, q, k, method, norm_descs)
            return dd, torch.where(ii >= 0, ii + shard_base, ii)
        return search_fn(db_shard, q, k, method, norm_descs, shard_base)
    equal = len(set(counts)) == 1 and counts[0] > 0
    if overlap == "auto":
        overlap = isinstance(db_shard, FlatIndex) and db_shard.has_planes and timings is None
    if overlap and equal and world > 1:
        # own queries against the shard while the others' rows travel; then the rest, in rank order around the own block
        own = counts[rank]
        qu_all = torch.empty(sum(counts), qu_local.shape[1], dtype=torch.float32, device=comm)
        src = qu_local.to(comm, torch.float32).contiguous()
        work = dist.all_gather_into_tensor(qu_all, src, group=group, async_op=True)
        d_own, i_own = run_search(qu_local)
        work.wait()
        if qu_all.device != qu_local.device:
            qu_all = qu_all.to(qu_local.device)
        lo, hi = rank * own, (rank + 1) * own
        rest = torch.cat([qu_all[:lo], qu_all[hi:]])
        d_r, i_r = run_search(rest)
        d = torch.cat([d_r[:lo], d_own.to(d_r.dtype), d_r[lo:]])
        i = torch.cat([i_r[:lo], i_own.to(i_r.dtype), i_r[lo:]])
    else:
        qu_all = gather_rows(qu_local, counts, rank, group, comm)
        if qu_all.device != qu_local.device:
            qu_all = qu_all.to(qu_local.device)
        t0 = lap("all_gather_ms", t0)
        d, i = run_search(qu_all)
    t0 = lap("search_ms", t0)
    # ONE gather for both lists: the fp32 distances ride as their bit patterns next to the int64 indices
    packed = torch.cat([d.to(torch.float32).contiguous().view(torch.int32).to(torch.int64), i.to(torch.int64)], dim=1).to(comm)
    out_p = [torch.empty_like(packed) for _ in range(world)] if rank == 0 else None
    dst = 0 if group is None or group is dist.group.WORLD else dist.get_global_rank(group, 0)
    dist.gather(packed, out_p, dst=dst, group=group)
    t0 = lap("gather_ms", t0)
    if rank != 0:
        return None, None
    kk = d.shape[1]
    host = [x.cpu() for x in out_p]
    res = merge_shard_topk([x[:, :kk].to(torch.int32).view(torch.float32).numpy() for x in host],
                           [x[:, kk:].numpy() for x in host], k, "ip" if method == "cosine" else "l2")
    lap("merge_ms", t0)
    return res
