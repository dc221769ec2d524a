"""``VLAD`` -- the reference's aggregation class (reference ``utilities.py:624-1008``;
distilled copy ``demo/utilities.py``) with every arithmetic step on the HIP
kernels of csrc/vlad.hip.  Same constructor, attributes, methods, prints,
asserts and cache-file protocol as the reference; CPU tensors in -> CPU tensors
out (the reference's callers pass ``ret.cpu()`` and call ``.numpy()`` on the
result), device tensors in -> device tensors out (additive fast path).
"""
import os
from typing import List, Union

import numpy as np
import torch

from . import _lib, ops
from .kmeans import KMeans

# compact <id>_t.pt (opt-in, cache_format = "lazy"): {"format": LAZY_FORMAT, "tokens": normalised tokens [N,D],
# "num_clusters": K}; the [N,K,D] residual tensor of the reference is tokens[:, None, :] - c_centers[None] and is only
# formed when generate_res_vec is asked for it.  It lives in its OWN file name so that a reference-side reader, which
# does torch.load(<id>_r.pt) and indexes it as a dense tensor (utilities.py:843-847, 951-954), never opens it.
LAZY_FORMAT = "anyloc_amd.lazy_residuals.v1"


def _as_tensor(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).to(torch.float32)
    return x


class VLAD:
    """
        An implementation of VLAD algorithm given database and query
        descriptors (constructor arguments as in the reference):
        num_clusters, desc_dim, intra_norm, norm_descs, dist_mode
        ('euclidean' | 'cosine'), vlad_mode ('soft' | 'hard'), soft_temp,
        cache_dir.
    """
    def __init__(self, num_clusters: int, desc_dim: Union[int, None] = None,
                 intra_norm: bool = True, norm_descs: bool = True, dist_mode: str = "cosine",
                 vlad_mode: str = "hard", soft_temp: float = 1.0,
                 cache_dir: Union[str, None] = None) -> None:
        self.num_clusters = num_clusters
        self.desc_dim = desc_dim
        self.intra_norm = intra_norm
        self.norm_descs = norm_descs
        self.mode = dist_mode
        self.vlad_mode = str(vlad_mode).lower()
        assert self.vlad_mode in ['soft', 'hard']
        self.soft_temp = soft_temp
        # Set in the training phase
        self.c_centers = None
        self.kmeans = None
        # Set the caching
        # how residual caches are WRITTEN: "reference" (default) = the reference's dense [N,K,D] tensor in <id>_r.pt,
        # byte-compatible with reference-side readers; "lazy" (opt-in, also ANYLOC_CACHE_FORMAT=lazy) = the compact
        # token file <id>_t.pt (3.25 MB instead of 104 MB per image at the headline shape).  Both are READ.
        self.cache_format = os.environ.get("ANYLOC_CACHE_FORMAT", "reference")
        if self.cache_format not in ("reference", "lazy"):
            raise ValueError(f"ANYLOC_CACHE_FORMAT must be 'reference' or 'lazy', got {self.cache_format!r}")
        self.cache_dir = cache_dir
        if self.cache_dir is not None:
            self.cache_dir = os.path.abspath(os.path.expanduser(self.cache_dir))
            if not os.path.exists(self.cache_dir):
                os.makedirs(self.cache_dir)
                print(f"Created cache directory: {self.cache_dir}")
            else:
                print("Warning: Cache directory already exists: " f"{self.cache_dir}")
        else:
            print("VLAD caching is disabled.")

    # ---------------------------------------------------------------- caches
    def can_use_cache_vlad(self):
        """True iff the cache directory exists and holds ``c_centers.pt``."""
        if self.cache_dir is None:
            return False
        if not os.path.exists(self.cache_dir):
            return False
        return os.path.exists(f"{self.cache_dir}/c_centers.pt")

    def can_use_cache_ids(self, cache_ids: Union[List[str], str, None],
                          only_residuals: bool = False) -> bool:
        """True iff every cache id has ``<id>_r.pt`` (or this package's compact ``<id>_t.pt``) and (unless
        ``only_residuals``) ``<id>_l.pt`` (hard) / ``<id>_s.pt`` (soft)."""
        if not self.can_use_cache_vlad():
            return False
        if cache_ids is None:
            return False
        if isinstance(cache_ids, str):
            cache_ids = [cache_ids]
        for cache_id in cache_ids:
            if self._residual_file(cache_id) is None:
                return False
            if self.vlad_mode == "hard" and not os.path.exists(
                    f"{self.cache_dir}/{cache_id}_l.pt") and not only_residuals:
                return False
            if self.vlad_mode == "soft" and not os.path.exists(
                    f"{self.cache_dir}/{cache_id}_s.pt") and not only_residuals:
                return False
        return True

    # ------------------------------------------------------------ vocabulary
    def fit(self, train_descs: Union[np.ndarray, torch.Tensor, None], process_group=None):
        """Build (or restore from ``cache_dir/c_centers.pt``) the cluster centres.

        ``process_group`` (additive, multi-GPU vocabulary build): ``train_descs`` is this rank's shard of
        the rows; every k-means iteration all-reduces the [K,D] sums and [K] counts, all ranks end with the same
        centres (identical to the flat fit on the concatenated rows) and rank 0 writes the cache."""
        self.kmeans = KMeans(self.num_clusters, mode=self.mode, process_group=process_group)
        if self.can_use_cache_vlad():
            print("Using cached cluster centers")
            self.c_centers = torch.load(f"{self.cache_dir}/c_centers.pt")
            self.kmeans.centroids = self.c_centers
            if self.desc_dim is None:
                self.desc_dim = self.c_centers.shape[1]
                print(f"Desc dim set to {self.desc_dim}")
        else:
            if train_descs is None:
                raise ValueError("No training descriptors given")
            train_descs = _as_tensor(train_descs)
            if self.desc_dim is None:
                self.desc_dim = train_descs.shape[1]
            home = train_descs.device
            x = ops._f32c(train_descs, _lib.require_gpu())
            if self.norm_descs:
                x = ops.l2norm_rows(x)
            self.kmeans.fit(x)
            self.kmeans.centroids = self.kmeans.centroids.to(home)
            self.c_centers = self.kmeans.centroids
            if self.cache_dir is not None:
                rank0 = True
                if process_group is not None:
                    import torch.distributed as dist
                    rank0 = dist.get_rank(process_group) == 0
                if rank0:
                    print("Caching cluster centers")
                    torch.save(self.c_centers, f"{self.cache_dir}/c_centers.pt")
                if process_group is not None:
                    dist.barrier(group=process_group)

    def fit_and_generate(self, train_descs: Union[np.ndarray, torch.Tensor]) -> torch.Tensor:
        """``fit`` on [num_imgs, num_descs, desc_dim] then VLADs of every image."""
        train_descs = _as_tensor(train_descs)
        self.fit(train_descs.reshape(-1, train_descs.shape[-1]))
        return self.generate_multi(train_descs)

    # ------------------------------------------------------------- generation
    def _check_fitted(self):
        assert self.kmeans is not None
        assert self.c_centers is not None

    def _centers_dev(self):
        dev = _lib.require_gpu()
        c = self.c_centers
        cached = getattr(self, "_c_dev", None)
        if cached is None or cached[0] is not c:
            self._c_dev = (c, ops._f32c(c, dev))
        return self._c_dev[1]

    def _cached_paths(self, cache_id):
        base = f"{self.cache_dir}/{cache_id}"
        return base + "_r.pt", base + "_l.pt", base + "_s.pt"

    def _residual_file(self, cache_id):
        """Path of the residual cache of ``cache_id``: the reference's ``<id>_r.pt`` first, else the compact
        ``<id>_t.pt``; None when neither exists."""
        for suffix in ("_r.pt", "_t.pt"):
            path = f"{self.cache_dir}/{cache_id}{suffix}"
            if os.path.isfile(path):
                return path
        return None

    def _load_residual_file(self, r_path):
        """residual cache file -> ("lazy", normalised tokens [N,D]) for the compact format (``<id>_t.pt``; round-2
        builds wrote the same dict under ``<id>_r.pt``), or ("dense", residual tensor [N,K,D]) for the reference's
        own format (utilities.py:963-970)."""
        obj = torch.load(r_path)
        if isinstance(obj, dict) and obj.get("format") == LAZY_FORMAT:
            return "lazy", obj["tokens"]
        return "dense", obj

    def _from_cache(self, cache_id):
        """Rebuild one VLAD on a cache hit (reference utilities.py:843-847 hard, :864-868 soft) WITHOUT the [N,K,D]
        residual tensor: the compact ``_r`` file holds the normalised tokens and the HIP kernel behind
        ``ops.vlad_assigned`` sums ``token - centre`` under the cached labels / soft weights.  A dense ``_r`` file
        written by the reference is honoured too: only the slice ``residuals[n, label_n]`` of every token is read
        (hard), i.e. N*D of its N*K*D values."""
        _, l_path, s_path = self._cached_paths(cache_id)
        kind, data = self._load_residual_file(self._residual_file(cache_id))
        K, D = self.num_clusters, self.desc_dim
        c = self._centers_dev()
        if self.vlad_mode == "hard":
            if not os.path.isfile(l_path):
                return None
            labels = torch.load(l_path).to(torch.int64)
            if kind == "lazy":
                out = ops.vlad_assigned(data, c, labels=labels, norm_descs=False, intra_norm=self.intra_norm)
            else:           # own-cluster residual of every token; summing it needs no centre any more
                picked = data[torch.arange(data.shape[0]), labels.cpu()].to(torch.float32)
                out = ops.vlad_assigned(picked, torch.zeros_like(c), labels=labels, norm_descs=False,
                                        intra_norm=self.intra_norm)
        else:
            if not os.path.isfile(s_path):
                return None
            soft = torch.load(s_path).to(torch.float32)
            if kind == "lazy":
                out = ops.vlad_assigned(data, c, soft=soft, norm_descs=False, intra_norm=self.intra_norm)
            else:
                # a dense file of the reference in soft mode: its quirk sums every cluster's residual (utilities.py:881-884),
                # so all N*K*D stored values are needed; reduced once over the cluster axis on the host, the rest on the
                # device (interchange path for files the reference wrote, not the hot path)
                dev = _lib.require_gpu()
                r_sum = data.to(dev).sum(1, dtype=torch.float64)                        # [N,D]
                un = (soft.to(dev).double().t() @ r_sum).to(torch.float32)
                if self.intra_norm:
                    un = ops.l2norm_rows(un)
                out = ops.l2norm_rows(un.reshape(1, K * D))[0]
        return out.cpu()

    def _write_cache(self, cache_id, descs_home):
        """Store what a later cache hit needs (reference utilities.py:850-852, :876-878, :963-970): ``_r`` + ``_l``
        (hard) / ``_s`` (soft).  By default ``_r`` is the reference's dense [N,K,D] tensor (written by the HIP residual
        kernel); ``cache_format == "lazy"`` writes the compact ``_t`` file instead -- the normalised tokens [N,D], from
        which ``residual[n,k] = token[n] - c_centers[k]`` is recomputed on demand."""
        r_path, l_path, s_path = self._cached_paths(cache_id)
        cid_dir = f"{self.cache_dir}/" f"{os.path.split(cache_id)[0]}"
        if not os.path.isdir(cid_dir):
            os.makedirs(cid_dir)
            print(f"Created directory: {cid_dir}")
        dev = _lib.require_gpu()
        x = ops._f32c(descs_home, dev)
        c = self._centers_dev()
        if self._residual_file(cache_id) is None:
            self._save_residuals(x, cache_id)
        if self.vlad_mode == "hard":
            if not os.path.isfile(l_path):
                torch.save(self.kmeans.predict(x).cpu(), l_path)
        elif not os.path.isfile(s_path):
            torch.save(ops.vlad_soft_weights(x, c, self.soft_temp).cpu(), s_path)

    def _save_residuals(self, x_dev, cache_id):
        if self.cache_format == "reference":
            torch.save(ops.vlad_residuals(x_dev, self._centers_dev(), self.norm_descs).cpu(),
                       f"{self.cache_dir}/{cache_id}_r.pt")
        else:
            xh = ops.l2norm_rows(x_dev) if self.norm_descs else x_dev
            torch.save({"format": LAZY_FORMAT, "tokens": xh.cpu(), "num_clusters": self.num_clusters},
                       f"{self.cache_dir}/{cache_id}_t.pt")

    # Descriptors that arrive as ONE CPU tensor (scripts/dino_v2_vlad.py:236-260 hands over [n_img, 529, 1536]: 3.25 MB per image)
    # go to the device in pieces: the copy runs at the PCIe rate either way (56 GB/s), but the device buffer of a piece is
    # allocated once and reused by the next piece, where ONE 832 MB buffer for 256 images cost the first call 33 ms of
    # allocation on top of the 15 ms copy.  64 MB since round 6 (256 MB in round 5): the first call of a process 23 - 40 ms and
    # every later one 22 ms on the probe's box, against 21 - 30 / 40 - 63 ms for 256 MB pieces and 35 - 53 / 33 for 16 MB
    # (tools/probe_host_staging.py, profiles/r06_host_staging.log); the result rows land in ONE tensor, slice by slice
    HOST_CHUNK_BYTES = 64 << 20

    def _generate_batch(self, multi_query):
        """[n_img,N,D] tensor or list of [N_i,D] -> [n_img, K*D] on the inputs' device."""
        first = multi_query[0] if not isinstance(multi_query, torch.Tensor) else multi_query
        first = _as_tensor(first)
        home = first.device
        if not isinstance(multi_query, torch.Tensor):
            multi_query = [_as_tensor(q) for q in multi_query]
        kw = dict(mode=self.vlad_mode, norm_descs=self.norm_descs, intra_norm=self.intra_norm,
                  soft_temp=self.soft_temp, dist_mode=self.mode)
        c = self._centers_dev()
        n = len(multi_query)
        step = n
        if home.type == "cpu" and isinstance(multi_query, torch.Tensor) and n > 1:
            per_img = max(1, multi_query[0].numel() * 4)
            step = max(1, self.HOST_CHUNK_BYTES // per_img)
        if step < n:
            # every piece with the workgroups-per-image count of the WHOLE batch: the bits of the one-call result
            if self.vlad_mode == "hard":
                kw["parts"] = ops.vlad_auto_parts(n, n * multi_query.shape[1], multi_query.shape[2], c.shape[0])
            out = torch.empty(n, c.shape[0] * c.shape[1], dtype=torch.float32, device=c.device)     # ONE result tensor, filled by slices
            for s in range(0, n, step):
                ops.vlad(multi_query[s:s + step], c, out=out[s:s + step], **kw)
        else:
            out = ops.vlad(multi_query, c, **kw)
        return ops.to_home(out, home)

    def generate(self, query_descs: Union[np.ndarray, torch.Tensor],
                 cache_id: Union[str, None] = None) -> torch.Tensor:
        """VLAD of one image: [n_q, desc_dim] -> [num_clusters * desc_dim]."""
        self._check_fitted()
        if self.desc_dim is None:
            self.desc_dim = self.c_centers.shape[1]
        if cache_id is not None and self.can_use_cache_vlad():
            if self._residual_file(cache_id) is not None:
                res = self._from_cache(cache_id)
                if res is not None:
                    return res
            if query_descs is not None:
                self._write_cache(cache_id, _as_tensor(query_descs))
        if query_descs is None:
            raise ValueError(f"no descriptors given and no usable cache for {cache_id!r}")
        return self._generate_batch(_as_tensor(query_descs)[None])[0]

    def generate_multi(self, multi_query: Union[np.ndarray, torch.Tensor, list],
                       cache_ids: Union[List[str], None] = None) -> Union[torch.Tensor, list]:
        """VLADs of several images ([n_imgs, n_kpts, d] tensor or a list of
        [n_kpts_i, d]); one batched launch when no cache ids are involved."""
        self._check_fitted()
        if cache_ids is None or all(c is None for c in cache_ids) or not self.can_use_cache_vlad():
            if len(multi_query) == 0:
                return torch.empty(0, self.num_clusters * (self.desc_dim or 0))
            if isinstance(multi_query, np.ndarray):
                multi_query = torch.from_numpy(multi_query).to(torch.float32)
            return self._generate_batch(multi_query)
        res = [self.generate(q, c) for (q, c) in zip(multi_query, cache_ids)]
        try:
            res = torch.stack(res)
        except TypeError:
            try:
                res = np.stack(res)
            except TypeError:
                pass
        return res

    def generate_res_vec(self, query_descs: Union[np.ndarray, torch.Tensor],
                         cache_id: Union[str, None] = None) -> torch.Tensor:
        """Residual tensor [n_q, n_c, d] = normalise(q)[:,None,:] - c_centers[None] (reference
        utilities.py:928-972).  The VLAD path of this class never forms it; this method does, on request, with the
        HIP kernel behind ``ops.vlad_residuals`` (a pure HBM write).  A cached ``<id>_r.pt`` / ``<id>_t.pt`` is honoured;
        a new cache entry is written in ``cache_format`` (see ``_write_cache``)."""
        self._check_fitted()
        if cache_id is not None and self.can_use_cache_vlad() and self._residual_file(cache_id) is not None:
            kind, data = self._load_residual_file(self._residual_file(cache_id))
            if kind == "dense":
                return data
            return ops.vlad_residuals(data, self._centers_dev(), norm_descs=False).to(data.device)
        query_descs = _as_tensor(query_descs)
        home = query_descs.device
        x = ops._f32c(query_descs, _lib.require_gpu())
        residuals = ops.vlad_residuals(x, self._centers_dev(), self.norm_descs)
        residuals = ops.to_home(residuals, home)
        if cache_id is not None and self.can_use_cache_vlad():
            cid_dir = f"{self.cache_dir}/" f"{os.path.split(cache_id)[0]}"
            if not os.path.isdir(cid_dir):
                os.makedirs(cid_dir)
                print(f"Created directory: {cid_dir}")
            self._save_residuals(x, cache_id)
        return residuals

    def generate_multi_res_vec(self, multi_query: Union[np.ndarray, torch.Tensor, list],
                               cache_ids: Union[List[str], None] = None) -> Union[torch.Tensor, list]:
        if cache_ids is None:
            cache_ids = [None] * len(multi_query)
        res = [self.generate_res_vec(q, c) for (q, c) in zip(multi_query, cache_ids)]
        try:
            res = torch.stack(res)
        except (TypeError, RuntimeError):
            try:
                res = np.stack(res)
            except (TypeError, ValueError):
                pass
        return res
