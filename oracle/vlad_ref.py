"""TEST INFRASTRUCTURE -- CPU restatement of the reference's VLAD aggregation
and recall computation (torch CPU fp32, the library the reference itself runs
these steps on).  Each function cites the reference lines it follows; the
restatements are pinned against the reference's own code (run verbatim through
``oracle/ref_loader.py``) by ``tests/golden/*.npz`` + ``tests/test_oracle_golden.py``.
"""
import numpy as np
import torch
from torch.nn import functional as F

from .faiss_flat import flat_search
from .fpk_kmeans import KMeans


def fpk_cosine_scores(x, centers):
    """fast-pytorch-kmeans ``cos_sim`` (rows / (norm + 1e-8)) -- the metric
    ``kmeans.predict`` uses at reference ``utilities.py:849``."""
    return KMeans.cos_sim(x, centers)


def hard_labels(x, centers):
    """``labels = self.kmeans.predict(query_descs)`` (``utilities.py:849``):
    argmax of the fpk cosine score of the tokens AS PASSED against the raw
    centroids.  torch.max returns the first maximal index on CPU."""
    return fpk_cosine_scores(x, centers).max(dim=-1)[1]


def vlad_hard(x, centers, norm_descs=True, intra_norm=True, labels=None):
    """Hard-assignment VLAD of one image.

    ``utilities.py:959-962``: residuals use the re-normalised tokens
    ``F.normalize(x)`` minus the RAW (un-normalised) centroids;
    ``:849``: labels from the tokens as passed; ``:854-861``: per used cluster,
    sum of the members' residuals w.r.t. that cluster, optional intra-norm,
    unused clusters stay zero; ``:889``: global L2 norm.
    x [N,D] float32, centers [K,D] float32 -> (vlad [K*D], labels [N] int64).
    """
    K, D = centers.shape
    if labels is None:
        labels = hard_labels(x, centers)
    xh = F.normalize(x) if norm_descs else x
    out = torch.zeros(K * D)
    for k in sorted(set(labels.tolist())):
        s = (xh[labels == k] - centers[k]).sum(dim=0)
        if intra_norm:
            s = F.normalize(s, dim=0)
        out[k * D:(k + 1) * D] = s
    return F.normalize(out, dim=0), labels


def vlad_soft(x, centers, soft_temp=1.0, norm_descs=True, intra_norm=True):
    """Soft-assignment VLAD with the reference's quirk (``utilities.py:870-887``):
    weights = softmax(temp * F.cosine_similarity(x, c)) on the tokens as
    passed; block k = sum over ALL tokens q AND ALL clusters c of
    ``w[q,k] * (xh[q] - centers[c])`` (the rearrange "(q c) d" at ``:883-884``
    sums every cluster's residual, not only cluster k's)."""
    K, D = centers.shape
    cos = F.cosine_similarity(x[:, None, :], centers[None, :, :], dim=2)
    w = F.softmax(soft_temp * cos, dim=1)                       # [N,K]
    xh = F.normalize(x) if norm_descs else x
    res = xh[:, None, :] - centers[None, :, :]                  # [N,K,D]
    out = torch.zeros(K * D)
    for k in range(K):
        s = (w[:, k, None, None] * res).reshape(-1, D).sum(dim=0)
        if intra_norm:
            s = F.normalize(s, dim=0)
        out[k * D:(k + 1) * D] = s
    return F.normalize(out, dim=0), w


def kmeans_fit(x, K, norm_descs=True, mode="cosine", init_idx=None):
    """``VLAD.fit`` without a cache (``utilities.py:779-787``): optional
    ``F.normalize`` of the rows, then fpk ``KMeans(K, mode).fit``.  ``init_idx``
    overrides the NumPy-global-RNG draw (tests pass the recorded draw so the
    HIP path and the oracle start from the same rows)."""
    if norm_descs:
        x = F.normalize(x)
    km = KMeans(K, mode=mode)
    km.fit(x, centroids=None if init_idx is None else x[torch.as_tensor(init_idx)])
    return km.centroids, km.n_iter_


def top_k_recall(top_k, db, qu, gt_pos, method="cosine", norm_descs=True,
                 use_percentage=True, sub_sample_db=1, sub_sample_qu=1):
    """``get_top_k_recall`` (``utilities.py:433-469``): normalise rows, flat
    IP / L2 search for max(top_k) neighbours, then for each query i and each k
    count a hit if any of ``idx[i,:k]*sub_sample_db`` is in
    ``gt_pos[i*sub_sample_qu]``; divide by #queries."""
    if qu.ndim == 1:
        qu = qu[None]
    if norm_descs:
        db, qu = F.normalize(db), F.normalize(qu)
    if method not in ("cosine", "l2"):
        raise NotImplementedError(f"Method: {method}")
    dist, idx = flat_search(qu, db, max(top_k), "ip" if method == "cosine" else "l2")
    recalls = recalls_from_indices(top_k, idx.numpy(), gt_pos, use_percentage,
                                   sub_sample_db, sub_sample_qu)
    return dist, idx, recalls


def recalls_from_indices(top_k, idx, gt_pos, use_percentage=True,
                         sub_sample_db=1, sub_sample_qu=1):
    recalls = {k: 0 for k in top_k}
    for i, retr in enumerate(idx):
        for k in top_k:
            if np.any(np.isin(retr[:k] * sub_sample_db, gt_pos[i * sub_sample_qu])):
                recalls[k] += 1
    if use_percentage:
        for k in recalls:
            recalls[k] /= len(idx)
    return recalls
