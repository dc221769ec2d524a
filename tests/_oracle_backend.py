"""TEST INFRASTRUCTURE: run the product's HOST logic on a GPU-less machine by replacing the
device entry points (anyloc_amd.ops / HipDinoV2) with the CPU oracle.  Used only by CPU tests
that drive the reference's unmodified scripts through our ``utilities`` surface; it proves the
plumbing (argument conventions, caching, return types), not the kernels."""
import torch
from torch.nn import functional as F

from oracle import dinov2_ref, faiss_flat, fpk_kmeans, pool_ref, vlad_ref


class OracleDinoV2:
    def __init__(self, name, state_dict, device, max_layer=None):
        have = 1 + max(int(k.split(".")[1]) for k in state_dict if k.startswith("blocks."))
        self.model = dinov2_ref.DinoVisionTransformer(name)
        self.model.blocks = self.model.blocks[:have]
        self.model.load_state_dict(state_dict, strict=True)
        self.model.eval()
        self.dim, self.depth = self.model.embed_dim, have
        self.device = torch.device("cpu")

    def eval(self):
        return self

    def to(self, device):
        return self

    @torch.no_grad()
    def __call__(self, img):
        return self.model(img.cpu().float())

    @torch.no_grad()
    def forward_taps(self, img, taps, use_cls=False, norm_taps=True, norm_concat=False):
        assert img.shape[-2] % 14 == 0 and img.shape[-1] % 14 == 0
        outs = [dinov2_ref.extract_facet(self.model, img.cpu().float(), l, f, use_cls, norm_taps) for l, f in taps]
        out = torch.cat(outs, dim=-1)
        return F.normalize(out, dim=-1) if norm_concat else out


def install(monkeypatch):
    from anyloc_amd import _lib, extractor, kmeans, ops
    cpu = torch.device("cpu")
    monkeypatch.setattr(_lib, "require_gpu", lambda: cpu)
    monkeypatch.setattr(ops, "l2norm_rows", lambda x, eps=1e-12, out=None: F.normalize(x.float(), dim=-1, eps=eps))

    def vlad(tokens, centers, mode="hard", norm_descs=True, intra_norm=True, soft_temp=1.0, return_labels=False,
             dist_mode="cosine", parts=0):
        parts = list(tokens) if not isinstance(tokens, torch.Tensor) else list(tokens if tokens.ndim == 3 else tokens[None])
        outs, labs = [], []
        for t in parts:
            t = torch.as_tensor(t).float()
            if mode == "hard":
                lab = None if dist_mode == "cosine" else fpk_kmeans.KMeans.euc_sim(t, centers.float()).max(dim=-1)[1]
                v, l = vlad_ref.vlad_hard(t, centers.float(), norm_descs, intra_norm, labels=lab)
                labs.append(l)
            else:
                v, _ = vlad_ref.vlad_soft(t, centers.float(), soft_temp, norm_descs, intra_norm)
            outs.append(v)
        out = torch.stack(outs) if outs else torch.empty(0, centers.numel())
        return (out, torch.cat(labs) if labs else None) if return_labels else out

    def kmeans_step(x, c, mode="cosine", want_labels=False):
        sim = fpk_kmeans.KMeans.cos_sim(x, c) if mode == "cosine" else fpk_kmeans.KMeans.euc_sim(x, c)
        lab = sim.max(dim=-1)[1]
        onehot = (lab[None, :] == torch.arange(c.shape[0])[:, None]).to(x.dtype)
        return onehot @ x, onehot.sum(-1), lab

    def kmeans_update(sums, counts, c_old):
        c_new = sums / counts[:, None]
        c_new[c_new != c_new] = 0
        return c_new, ((c_new - c_old) ** 2).sum().double().reshape(1)

    def pool(tokens, method="average", gem_p=3.0):
        if method not in ops.POOL_MODES:
            raise NotImplementedError(f"ID: {method}")
        parts = list(tokens) if not isinstance(tokens, torch.Tensor) else list(tokens if tokens.ndim == 3 else tokens[None])
        outs = []
        for t in parts:
            t = torch.as_tensor(t).float()[None]
            outs.append(pool_ref.global_pool(t, method)[0] if method in ("average", "avg", "max")
                        else pool_ref.gem_descriptors(t, gem_p, method == "gem_abs")[0])
        return torch.stack(outs)

    def vlad_residuals(tokens, centers, norm_descs=True):
        x = tokens.float()
        return (F.normalize(x, dim=-1) if norm_descs else x)[:, None, :] - centers.float()[None]

    def vlad_assigned(tokens, centers, labels=None, soft=None, norm_descs=True, intra_norm=True):
        x, c = tokens.float(), centers.float()
        xh = F.normalize(x, dim=-1) if norm_descs else x
        K, D = c.shape
        if labels is not None:
            out = torch.zeros(K, D)
            for k in sorted(set(labels.tolist())):
                out[k] = (xh[labels == k] - c[k]).sum(0)
        else:
            res = xh[:, None, :] - c[None]
            out = torch.stack([(soft[:, k, None, None] * res).reshape(-1, D).sum(0) for k in range(K)])
        if intra_norm:
            out = F.normalize(out, dim=1)
        return F.normalize(out.reshape(-1), dim=0)

    def vlad_soft_weights(tokens, centers, soft_temp=1.0):
        cos = F.cosine_similarity(tokens.float()[:, None, :], centers.float()[None], dim=2)
        return F.softmax(soft_temp * cos, dim=1)

    monkeypatch.setattr(ops, "vlad_residuals", vlad_residuals)
    monkeypatch.setattr(ops, "vlad_assigned", vlad_assigned)
    monkeypatch.setattr(ops, "vlad_soft_weights", vlad_soft_weights)
    monkeypatch.setattr(ops, "pool", pool)
    monkeypatch.setattr(ops, "gemm_nt", lambda a, w, bias=None: a.float() @ w.float().t() + (0 if bias is None else bias))
    monkeypatch.setattr(ops, "_f32c", lambda t, device=None: t.detach().to("cpu", torch.float32).contiguous())

    def pca_gram_f64(x, mean64, side):                       # csrc/pca_f64.hip: the fit's symmetric matrix in float64
        xw = x.double() - mean64.double()
        return xw @ xw.t() if side == 0 else xw.t() @ xw

    monkeypatch.setattr(ops, "pca_gram_f64", pca_gram_f64)
    monkeypatch.setattr(ops, "pca_axes_f64", lambda vec, k, x, mean64: vec[:, :k].double().t() @ (x.double() - mean64.double()))
    monkeypatch.setattr(ops, "vlad", vlad)
    monkeypatch.setattr(ops, "vlad_auto_parts", lambda n_img, n_tok, D, K: 1)
    monkeypatch.setattr(ops, "kmeans_step", kmeans_step)
    monkeypatch.setattr(ops, "kmeans_update", kmeans_update)
    monkeypatch.setattr(kmeans, "_local_step", kmeans_step)
    monkeypatch.setattr(ops, "topk", lambda q, db, k, metric="ip", index_base=0, normalize_db=False:
                        faiss_flat.flat_search(q.float(), F.normalize(db.float(), dim=-1) if normalize_db else db.float(),
                                               k, metric))
    monkeypatch.setattr(extractor, "HipDinoV2", OracleDinoV2)
