"""TEST INFRASTRUCTURE -- CPU restatement of fast-pytorch-kmeans==0.1.6 ``KMeans``.

The reference pins ``fast-pytorch-kmeans==0.1.6`` (reference
``requirements.txt:54``, ``setup_conda.sh:224``) and calls it from
``utilities.py:766`` (ctor), ``:772`` (centroids assigned), ``:786-787``
(fit / centroids read) and ``:849`` (predict).  The package is not vendored in
``/root/reference`` and not installed here, so its published algorithm is
restated.  Only the CPU branch is restated: the GPU
branch of fpk differs only in memory-driven sub-batching of ``max_sim``.

Semantics that matter for parity (each exercised by tests/test_oracle_*.py):
  * cosine similarity divides rows by ``(norm + 1e-8)`` (NOT F.normalize's
    ``max(norm, eps)``),
  * init centroids = ``X[np.random.choice(N, K, replace=False)]`` drawn from the
    NumPy *global* RNG (seeded 42 by the reference's ``seed_everything``),
  * full-batch update, means are NOT re-normalised in cosine mode,
  * an empty cluster's mean is 0/0 = NaN -> replaced by 0,
  * stop when ``sum((c_new - c)**2) <= tol`` (1e-4) or after 100 iterations.
"""
import numpy as np
import torch


class KMeans:
    def __init__(self, n_clusters, max_iter=100, tol=1e-4, verbose=0,
                 mode="euclidean", minibatch=None):
        self.n_clusters = n_clusters
        self.max_iter = max_iter
        self.tol = tol
        self.verbose = verbose
        self.mode = mode
        self.minibatch = minibatch
        self.centroids = None
        self.n_iter_ = 0          # not in fpk; recorded for tests

    # -- similarity ---------------------------------------------------
    @staticmethod
    def cos_sim(a, b):
        a_n = a / (a.norm(dim=-1, keepdim=True) + 1e-8)
        b_n = b / (b.norm(dim=-1, keepdim=True) + 1e-8)
        return a_n @ b_n.transpose(-2, -1)

    @staticmethod
    def euc_sim(a, b):
        return (2 * a @ b.transpose(-2, -1)
                - (a ** 2).sum(dim=1)[..., :, None]
                - (b ** 2).sum(dim=1)[..., None, :])

    def max_sim(self, a, b):
        sim = self.cos_sim(a, b) if self.mode == "cosine" else self.euc_sim(a, b)
        return sim.max(dim=-1)

    # -- fitting ------------------------------------------------------
    def fit_predict(self, X, centroids=None):
        assert isinstance(X, torch.Tensor), "fpk takes torch tensors"
        n, _ = X.shape
        if centroids is None:
            pick = np.random.choice(n, size=[self.n_clusters], replace=False)
            self.centroids = X[pick]
        else:
            self.centroids = centroids
        closest = None
        for it in range(self.max_iter):
            closest = self.max_sim(X, self.centroids)[1]
            onehot = (closest[None, :] ==
                      torch.arange(self.n_clusters)[:, None]).to(X.dtype)
            c_new = (onehot @ X) / onehot.sum(-1)[:, None]
            c_new[c_new != c_new] = 0          # empty cluster: NaN -> 0
            err = ((c_new - self.centroids) ** 2).sum()
            self.centroids = c_new              # lr == 1 for full batch
            self.n_iter_ = it + 1
            if err <= self.tol:
                break
        return closest

    def fit(self, X, centroids=None):
        self.fit_predict(X, centroids)

    def predict(self, X):
        assert isinstance(X, torch.Tensor), "fpk takes torch tensors"
        return self.max_sim(X, self.centroids)[1]
