"""PCA (optionally whitened) projection of global descriptors on the device.

The reference reduces its 49 152-d VLADs with ``sklearn.decomposition.PCA(lower_dim, svd_solver='full',
whiten=...)`` on the host (``utilities.py:522-586``, called at ``scripts/dino_v2_vlad.py:357-369``; the
joint variant at ``scripts/joint_pca_project.py:84-86``): a LAPACK SVD of the centred [n, f] matrix.

Here the decomposition is reduced to GEMMs + one symmetric eigenproblem on the device, and the projections
run on the fp32 MFMA GEMM of the library (``anyloc_gemm_nt``):

* fit:  centre on the device; the smaller of the Gram matrix  Xc Xc^T [n, n]  and the scatter matrix
  Xc^T Xc [f, f]  is one GEMM (in float64 on the double-precision matrix cores by default -- ``anyloc_pca_gram_f64``,
  csrc/pca_f64.hip -- see ``precise``); its symmetric eigendecomposition (float64, ``torch.linalg.eigh`` -- a
  dense-solver library call, the one step of the fit that is not a kernel of this package) gives the singular values
  and one side of the SVD; for the Gram side the principal axes follow from one more GEMM,
  V^T = diag(1/s) U^T Xc (``anyloc_pca_axes_f64``).
  Signs follow sklearn's ``svd_flip``: by default the rule of the sklearn installed next to this package (what the
  reference's own call would produce here): u-based (largest-magnitude entry of every U column positive) for the
  versions the reference pins, v-based from sklearn 1.5 on; ``sign_convention="u"`` / ``"v"`` force one.
* transform:  one GEMM with the bias epilogue,  X W^T - W mean,  W = components (/ sqrt(explained variance)
  when whitening) -- the order sklearn's ``_BasePCA._transform`` uses.

Attributes mirror sklearn's (``mean_, components_, explained_variance_, singular_values_, n_components_``)
as device tensors.  
"""
import numpy as np
import torch

from . import _lib, ops


def _gemm(a, w, bias=None):
    """a [M,K] w[N,K]^T (+ bias) on the library GEMM; K zero-padded to the multiple of 4 the kernel's
    16-byte loads need (zeros add nothing to the sums)."""
    pad = -a.shape[1] % 4
    if pad:
        a, w = torch.nn.functional.pad(a, (0, pad)), torch.nn.functional.pad(w, (0, pad))
    return ops.gemm_nt(a, w, bias)


def default_sign_convention():
    """The ``svd_flip`` rule of the sklearn the reference would import here: u-based before 1.5 (incl. the pinned
    0.24.2 / 1.0.2), v-based from 1.5 on; "u" when sklearn is not installed."""
    try:
        import sklearn
        major, minor = (int(p) for p in sklearn.__version__.split(".")[:2])
        return "v" if (major, minor) >= (1, 5) else "u"
    except Exception:
        return "u"


class PCA:
    def __init__(self, n_components: int, whiten: bool = False, precise: bool = True, sign_convention: str = "auto"):
        """``sign_convention``: which factor of the SVD fixes the sign of every axis -- "u": the largest-magnitude entry
        of each column of U is positive, the ``svd_flip`` of sklearn < 1.5 incl. the 0.24.2 / 1.0.2 the reference pins
        (``setup_conda.sh:234``); "v": the largest-magnitude entry of each principal axis is positive, sklearn >= 1.5
        (``svd_flip(u, vt, u_based_decision=False)``); "auto" (default): what the reference's own
        ``sklearn.decomposition.PCA`` call would do in THIS environment -- the rule of the installed sklearn, "u" when
        none is installed -- so dumps compare element-wise with the reference run side by side, in its pinned
        environment as well as in a current one.  Projected coordinates differ by a per-axis sign between the two rules;
        retrieval is unaffected."""
        if sign_convention == "auto":
            sign_convention = default_sign_convention()
        if sign_convention not in ("v", "u"):
            raise ValueError("sign_convention must be 'auto', 'v' or 'u'")
        self.sign_convention = sign_convention
        self.n_components = int(n_components)
        self.whiten = bool(whiten)
        # The Gram / scatter matrix squares the condition number: formed with fp32 sums, the axes whose variance is
        # below ~1e-6 x the largest are noise (measured on the MI355X: axis 64 of a 0.9^k spectrum off by 1e-3).
        # `precise` (default) forms that one symmetric matrix, and the k x f back-projection, in float64 on
        # v_mfma_f64_16x16x4_f64 (csrc/pca_f64.hip), which brings the fit to the accuracy of the reference's float32
        # LAPACK SVD or better; precise=False keeps everything on the fp32 MFMA kernel (fine for leading axes).
        self.precise = bool(precise)

    # ---------------------------------------------------------------- fit ----
    def fit(self, X):
        device = _lib.require_gpu()
        X = ops._f32c(torch.as_tensor(X), device)
        if X.dim() != 2:
            raise ValueError(f"Expected 2D array, got {X.dim()}D array instead")
        n, f = X.shape
        k = self.n_components
        if not 1 <= k <= min(n, f):
            raise ValueError(f"n_components={k} must be between 1 and min(n_samples, n_features)={min(n, f)} "
                             f"with svd_solver='full'")
        mean64 = torch.zeros(f, dtype=torch.float64, device=device)
        rows = max(1, (64 << 20) // (8 * f))                   # float64 column sums over row blocks: bounded temporaries
        for r0 in range(0, n, rows):
            mean64 += X[r0:r0 + rows].sum(dim=0, dtype=torch.float64)
        mean64 /= n
        self.mean_ = mean64.to(torch.float32)
        # precise: the symmetric matrix and the back-projection are formed in float64 by the library's own kernel on the
        # double-precision matrix cores (csrc/pca_f64.hip), which centres the fp32 data with the float64 mean on the way into
        # LDS -- neither a float64 nor a centred fp32 copy of X is made (10 000 x 49 152: 3.9 + 2.0 GB)
        xc = None if self.precise else X - self.mean_
        if n <= f:
            gram = ops.pca_gram_f64(X, mean64, 0) if self.precise else _gemm(xc, xc)         # [n, n] = Xc Xc^T
            lam, vec = self._eigh_desc(gram)
            s = lam.clamp_min(0).sqrt()
            if self.precise:
                axes = ops.pca_axes_f64(vec, k, X, mean64) / s[:k].clamp_min(1e-300)[:, None]  # [k, f] = U^T Xc / s
                axes = axes.to(torch.float32)
            else:
                u_t = vec[:, :k].t().to(torch.float32).contiguous()            # [k, n]
                axes = _gemm(u_t, xc.t().contiguous())
                axes = axes / s[:k].to(torch.float32).clamp_min(torch.finfo(torch.float32).tiny)[:, None]
        else:
            if self.precise:
                scatter = ops.pca_gram_f64(X, mean64, 1)                   # [f, f] = Xc^T Xc
            else:
                xt = xc.t().contiguous()
                scatter = _gemm(xt, xt)
            lam, vec = self._eigh_desc(scatter)
            s = lam.clamp_min(0).sqrt()
            axes = vec[:, :k].t().to(torch.float32).contiguous()
        # axes of (numerically) zero variance -- rank-deficient data, e.g. n_components == n_samples after centring -- carry
        # no direction: U^T Xc / s would divide noise by ~0.  They are set to zero (every projection onto them is 0,
        # also under whitening) instead of being blown up to inf / NaN.
        # Threshold = the singular values the decomposition's own rounding noise reaches.  `precise`: the symmetric matrix is
        # formed and decomposed in float64, its eigenvalues are good to ~ max(n, f) eps64 lambda_0, i.e. singular values to
        # sqrt(4 max(n, f) eps64) s_0 ~ 1e-6 s_0 (the Gram route squares the condition number) -- small but real components
        # of the data above that are kept, as sklearn keeps them.  All-fp32 path: eps32 sqrt(max(n, f)) s_0 of the fp32 data.
        big = float(max(n, f))
        if self.precise:
            dead = s[:k] <= s[0] * (4.0 * big * torch.finfo(torch.float64).eps) ** 0.5
        else:
            dead = s[:k] <= s[0] * 4.0 * big ** 0.5 * torch.finfo(torch.float32).eps
        axes = torch.where(dead[:, None].to(axes.device), torch.zeros_like(axes), axes)
        # unit length in fp32 (the GEMM above leaves ~1e-7 of drift) and sklearn's sign rule
        axes = torch.nn.functional.normalize(axes, dim=1)
        if self.sign_convention == "v":
            piv = axes.abs().argmax(dim=1)
            sign = torch.sign(axes[torch.arange(k, device=device), piv])
        else:
            cols = torch.arange(k, device=device)
            if xc is None and n <= f:
                # Gram side: U is in hand -- the eigenvectors the axes were back-projected from (axes = U^T Xc / s and the
                # normalisation scale rows by positive numbers, so U's largest entry carries the sign sklearn looks at)
                uk = vec[:, :k]
                sign = torch.sign(uk[uk.abs().argmax(dim=0), cols]).to(torch.float32)
            elif xc is None:
                # scatter side: U diag(s) = Xc V, formed from row blocks CENTRED FIRST (round 5 took X V + (-mean V) as a bias:
                # with |mean v_k| far above the column's largest |u| entry the fp32 cancellation could pick the wrong pivot)
                best = torch.zeros(k, device=device)
                sign = torch.ones(k, device=device)
                at = axes.contiguous()
                rows_b = max(1, (64 << 20) // (4 * f))
                for r0 in range(0, n, rows_b):
                    ub = _gemm((X[r0:r0 + rows_b] - self.mean_).contiguous(), at)
                    pb = ub.abs().argmax(dim=0)
                    vb = ub[pb, cols]
                    take = vb.abs() > best
                    best = torch.where(take, vb.abs(), best)
                    sign = torch.where(take, torch.sign(vb), sign)
            else:
                u = _gemm(xc, axes.contiguous())                           # [n, k] = U diag(s): same signs as U
                sign = torch.sign(u[u.abs().argmax(dim=0), cols])
        sign = torch.where(sign == 0, torch.ones_like(sign), sign)
        self.components_ = (axes * sign[:, None]).contiguous()
        self._dead = dead
        self.singular_values_ = s[:k].to(torch.float32)
        self.explained_variance_ = (lam[:k].clamp_min(0) / max(n - 1, 1)).to(torch.float32)
        total = float(lam.clamp_min(0).sum() / max(n - 1, 1))
        self.explained_variance_ratio_ = self.explained_variance_ / total if total > 0 else self.explained_variance_
        self.n_components_, self.n_samples_, self.n_features_in_ = k, n, f
        self._w = None
        return self

    @staticmethod
    def _eigh_desc(sym):
        lam, vec = torch.linalg.eigh(sym.to(torch.float64))
        return lam.flip(0), vec.flip(1)

    # ---------------------------------------------------------- transform ----
    def _projection(self):
        if self._w is None:
            w = self.components_
            if self.whiten:
                scale = self.explained_variance_.sqrt().clamp_min(torch.finfo(torch.float32).eps)
                w = w / scale[:, None]                                      # (dead axes are zero rows: 0 / eps = 0)
            w = w.contiguous()
            self._w = (w, -(w.double() * self.mean_.double()).sum(dim=1).to(torch.float32))
        return self._w

    def transform(self, X):
        if not hasattr(self, "components_"):
            raise RuntimeError("This PCA instance is not fitted yet. Call 'fit' first.")
        X = torch.as_tensor(X)
        src = X.device
        x = ops._f32c(X, _lib.require_gpu())
        if x.dim() != 2 or x.shape[1] != self.n_features_in_:
            raise ValueError(f"X has {x.shape[-1]} features, but PCA is expecting {self.n_features_in_} features")
        w, b = self._projection()
        out = _gemm(x, w, b)
        return out if src.type == "cuda" else out.to(src)

    def fit_transform(self, X):
        return self.fit(X).transform(X)


def joint_pca_project(db_descs, qu_descs, lower_dim: int = 512, whiten: bool = False, sign_convention: str = "auto"):
    """Joint PCA projection of several datasets' global descriptors (reference ``scripts/joint_pca_project.py:62-101``):
    ONE PCA is fitted on the concatenation of all database descriptor sets, and every database / query set is projected
    with it.  ``db_descs`` / ``qu_descs``: lists of [n_i, f] tensors (or arrays), one entry per dataset, as the script
    loads them from ``db-<name>.pt`` / ``qu-<name>.pt``.  Returns (list of projected database sets, list of projected
    query sets, the fitted PCA), each set on the device / in the container type it came in."""
    as_np = [isinstance(d, np.ndarray) for d in db_descs]
    dbs = [torch.as_tensor(d) for d in db_descs]
    qus = [torch.as_tensor(q) for q in qu_descs]
    if len(dbs) != len(qus) or not dbs:
        raise ValueError("need one database and one query descriptor set per dataset")
    pca = PCA(lower_dim, whiten=whiten, sign_convention=sign_convention)
    dev = _lib.require_gpu()
    down_db = pca.fit_transform(torch.cat([ops._f32c(d, dev) for d in dbs], dim=0))
    down_qu = pca.transform(torch.cat([ops._f32c(q, dev) for q in qus], dim=0))
    out_db, out_qu, i_db, i_qu = [], [], 0, 0
    for d, q, np_in in zip(dbs, qus, as_np):
        a, b = down_db[i_db:i_db + d.shape[0]], down_qu[i_qu:i_qu + q.shape[0]]
        i_db, i_qu = i_db + d.shape[0], i_qu + q.shape[0]
        if np_in:
            a, b = a.cpu().numpy(), b.cpu().numpy()
        else:
            a, b = a.to(d.device), b.to(q.device)
        out_db.append(a)
        out_qu.append(b)
    return out_db, out_qu, pca


def reduce_pca(train_descs, test_descs, lower_dim: int, low_factor: float = 0.0, fallback: int = 256,
               svd_solver: str = "full", whitening: bool = False):
    """Device version of reference ``utilities.py:522-586`` (same arguments and return convention: numpy in ->
    numpy out; tensors in -> tensors on the input's device).  ``svd_solver`` is accepted for signature
    compatibility: the decomposition is always the exact (full) one."""
    assert 0 <= low_factor <= 1
    as_np = isinstance(train_descs, np.ndarray)
    tr, ts = torch.as_tensor(train_descs), torch.as_tensor(test_descs)
    if low_factor == 0.0:
        pca = PCA(lower_dim, whiten=whitening)
        out_tr, out_ts = pca.fit_transform(tr), pca.transform(ts)
    else:
        n_samples, n_components = tr.shape
        if n_samples < n_components:
            print(f"Too few samples, fallback to {fallback}d first")
            both = PCA(fallback).fit_transform(torch.cat((tr, ts)))
            tr, ts = both[:n_samples], both[n_samples:]
        n_down = int(low_factor * lower_dim)
        n_up = lower_dim - n_down
        print(f"Up: {n_up}, Down: {n_down}")
        full = PCA(tr.shape[1]).fit(tr)
        basis = torch.cat((full.components_[:n_up], full.components_[-n_down:]))
        sel = PCA(lower_dim)
        sel.mean_, sel.components_, sel.n_features_in_, sel._w = full.mean_, basis.contiguous(), tr.shape[1], None
        sel.explained_variance_ = torch.cat((full.explained_variance_[:n_up], full.explained_variance_[-n_down:]))
        out_tr, out_ts = sel.transform(tr), sel.transform(ts)
    if as_np:
        return out_tr.cpu().numpy(), out_ts.cpu().numpy()
    return out_tr, out_ts
