"""TEST INFRASTRUCTURE -- CPU restatement of faiss 1.7.2 flat indexes.

The reference pins ``faiss-gpu==1.7.2`` (reference ``requirements.txt:53``,
``setup_conda.sh:222``) and uses it only at ``utilities.py:440-450``:
``IndexFlatIP(D)`` / ``IndexFlatL2(D)``, ``.add(db)``, ``.search(qu, k)`` on
torch tensors (``faiss.contrib.torch_utils`` makes add/search accept and return
torch tensors, ``utilities.py:13-14``).  faiss is not vendored and not
installed, so the published semantics are restated: exact brute force, results
best-first (IP: descending inner product; L2: ascending *squared* distance),
``D`` float32 ``[n,k]``, ``I`` int64 ``[n,k]``, ``k > ntotal`` pads ``I`` with
-1 (and ``D`` with -inf / +inf).  faiss leaves the order of exactly tied
scores unspecified; this restatement (and the HIP kernel) break ties towards
the lower database index.
"""
import numpy as np
import torch

METRIC_INNER_PRODUCT = 0
METRIC_L2 = 1


def _as_tensor(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), True
    return x.detach().to("cpu", torch.float32).contiguous(), False


def flat_search(qu, db, k, metric="ip", chunk=4096):
    """Exact top-k with lower-index tie break.  qu [Q,D], db [N,D] float32."""
    Q, N = qu.shape[0], db.shape[0]
    kk = min(k, N)
    dist = torch.empty(Q, k, dtype=torch.float32)
    idx = torch.full((Q, k), -1, dtype=torch.int64)
    dist.fill_(float("-inf") if metric == "ip" else float("inf"))
    if kk == 0 or Q == 0:
        return dist, idx
    for s in range(0, Q, chunk):
        q = qu[s:s + chunk]
        if metric == "ip":
            score = q @ db.T
        else:
            score = ((q * q).sum(1)[:, None] + (db * db).sum(1)[None, :]
                     - 2.0 * (q @ db.T))
            score = -score
        # stable sort descending => equal scores keep ascending index order
        order = torch.sort(score, dim=1, descending=True, stable=True)[1][:, :kk]
        best = torch.gather(score, 1, order)
        dist[s:s + chunk, :kk] = best if metric == "ip" else -best
        idx[s:s + chunk, :kk] = order
    return dist, idx


class _IndexFlat:
    metric = "ip"

    def __init__(self, d):
        self.d = int(d)
        self.ntotal = 0
        self._db = torch.empty(0, self.d, dtype=torch.float32)

    def add(self, x):
        x, _ = _as_tensor(x)
        assert x.ndim == 2 and x.shape[1] == self.d
        self._db = torch.cat([self._db, x], 0)
        self.ntotal = self._db.shape[0]

    def search(self, x, k):
        xt, was_np = _as_tensor(x)
        assert xt.ndim == 2 and xt.shape[1] == self.d
        D, I = flat_search(xt, self._db, int(k), self.metric)
        if was_np:
            return D.numpy(), I.numpy()
        return D, I

    def reset(self):
        self.__init__(self.d)


class IndexFlatIP(_IndexFlat):
    metric = "ip"


class IndexFlatL2(_IndexFlat):
    metric = "l2"


class StandardGpuResources:                      # utilities.py:446 (unused path)
    pass


def index_cpu_to_gpu(res, dev, index):            # utilities.py:447 (unused path)
    return index
