"""``KMeans`` with the fast-pytorch-kmeans 0.1.6 surface the reference touches
(``fpk.KMeans(K, mode=...)``, ``.fit``, ``.predict``, ``.centroids`` read AND
assigned -- reference ``utilities.py:766,772,786-787,849``), iterating on the HIP
assign+accumulate kernel (csrc/vlad.hip, ``anyloc_kmeans_step``).

Host logic mirrors fpk's ``fit_predict``: init rows drawn with
``np.random.choice(N, K, replace=False)`` from NumPy's *global* RNG, full-batch
update ``c = sums / counts`` with empty clusters -> 0 (fpk: NaN -> 0), means not
re-normalised, stop when ``sum((c_new - c)^2) <= tol`` or after ``max_iter``.

Multi-GPU: rows are sharded over ranks; pass ``process_group`` and
each iteration all-reduces the [K,D] sums and [K] counts (RCCL over xGMI),
every rank then computes identical centroids.
"""
import numpy as np
import torch

from . import _lib, ops


def _local_step(x, centroids, mode, want_labels=False):
    return ops.kmeans_step(x, centroids, mode, want_labels)


class KMeans:
    def __init__(self, n_clusters, max_iter=100, tol=1e-4, verbose=0, mode="euclidean",
                 minibatch=None, process_group=None, step_fn=None):
        if mode not in ("cosine", "euclidean"):
            raise NotImplementedError(f"mode {mode!r}")
        if minibatch is not None:
            raise NotImplementedError("minibatch k-means is not used by the reference path")
        self.n_clusters = n_clusters
        self.max_iter = max_iter
        self.tol = tol
        self.verbose = verbose
        self.mode = mode
        self.minibatch = minibatch
        self.centroids = None
        self.n_iter_ = 0
        self.process_group = process_group
        self._step = step_fn or _local_step       # injectable for CPU tests of the host logic

    # -- helpers ------------------------------------------------------------
    def _to_dev(self, t):
        if isinstance(t, np.ndarray):
            t = torch.from_numpy(t)
        if self._step is _local_step:
            return ops._f32c(t, _lib.require_gpu())
        return t.to(torch.float32)

    def _comm_device(self, like):
        """Device the collectives of ``process_group`` move: the tensors' own GPU for RCCL (backend "nccl" rejects CPU
        tensors), the host for a gloo group (CPU tests, single-GPU tests)."""
        import torch.distributed as dist
        if dist.get_backend(self.process_group) == "gloo":
            return torch.device("cpu")
        return like.device

    def _all_reduce(self, *tensors):
        """Sum over the ranks of ``process_group`` in place (RCCL; gloo groups are staged through the host, gloo has
        no device all-reduce for every dtype / build)."""
        if self.process_group is None:
            return
        import torch.distributed as dist
        for t in tensors:
            comm = self._comm_device(t)
            if comm != t.device:
                c = t.to(comm)
                dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.process_group)
                t.copy_(c)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group)

    def _sharded_init(self, x):
        """Initial centroids of a row-sharded fit, identical to the flat fit on the concatenated rows: rank 0 draws
        ``np.random.choice(N_total, K, replace=False)`` from NumPy's global RNG (what fpk does on the whole array),
        the draw is broadcast, every rank contributes the drawn rows it owns and the [K, D] table is all-reduced.
        Every tensor handed to a collective lives on the group's comm device (``_comm_device``)."""
        import torch.distributed as dist
        g = self.process_group
        world, rank = dist.get_world_size(g), dist.get_rank(g)
        comm = self._comm_device(x)
        counts = torch.zeros(world, dtype=torch.int64, device=comm)
        dist.all_gather_into_tensor(counts, torch.tensor([x.shape[0]], dtype=torch.int64, device=comm), group=g)
        counts = [int(c) for c in counts.cpu()]
        n_total, off = sum(counts), sum(counts[:rank])
        pick = torch.zeros(self.n_clusters, dtype=torch.int64, device=comm)
        if rank == 0:
            pick = torch.from_numpy(np.random.choice(n_total, size=[self.n_clusters], replace=False)).to(torch.int64).to(comm)
        dist.broadcast(pick, src=dist.get_global_rank(g, 0) if g is not dist.group.WORLD else 0, group=g)
        pick = pick.to(x.device)
        c = torch.zeros(self.n_clusters, x.shape[1], dtype=torch.float32, device=x.device)
        mine = (pick >= off) & (pick < off + x.shape[0])
        if bool(mine.any()):
            c[mine] = x[pick[mine] - off]
        self._all_reduce(c)
        return c

    # -- fpk surface -----------------------------------------------------------
    def fit_predict(self, X, centroids=None):
        home = X.device if isinstance(X, torch.Tensor) else torch.device("cpu")
        x = self._to_dev(X)
        n = x.shape[0]
        if centroids is None:
            if self.process_group is not None:
                c = self._sharded_init(x)
            else:
                pick = np.random.choice(n, size=[self.n_clusters], replace=False)
                c = x[torch.as_tensor(pick, device=x.device)].clone()
        else:
            c = self._to_dev(centroids).clone()
        labels = None
        for it in range(self.max_iter):
            sums, counts, labels = self._step(x, c, self.mode, True)
            self._all_reduce(sums, counts)
            if self._step is _local_step:
                # one launch: division, empty cluster -> 0, convergence error (float64); 8 bytes come back
                c_new, err = ops.kmeans_update(sums, counts, c)
            else:                                     # injected step (CPU tests of the host logic)
                c_new = sums / counts[:, None]
                c_new = torch.where(counts[:, None] > 0, c_new, torch.zeros_like(c_new))
                err = ((c_new - c) ** 2).sum()
            c = c_new
            self.n_iter_ = it + 1
            if self.verbose:
                print(f"iter {it}: error {float(err):.3e}")
            if float(err) <= self.tol:
                break
        self.centroids = c.to(home)
        return labels.to(home)

    def fit(self, X, centroids=None):
        self.fit_predict(X, centroids)

    def predict(self, X):
        if self.centroids is None:
            raise RuntimeError("KMeans.predict before fit / before centroids were assigned")
        home = X.device if isinstance(X, torch.Tensor) else torch.device("cpu")
        x = self._to_dev(X)
        if x.shape[0] == 0:
            return torch.empty(0, dtype=torch.int64, device=home)
        _, _, labels = self._step(x, self._to_dev(self.centroids), self.mode, True)
        return labels.to(home)
