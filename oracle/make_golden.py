"""TEST INFRASTRUCTURE -- record golden vectors by running the REFERENCE's own code.

Run in the build container only (needs ``/root/reference``):

    cd /root/repo && python -m oracle.make_golden

Everything arithmetic below is executed by the reference's unmodified
``utilities.py`` (``DinoV2ExtractFeatures.__call__``, ``VLAD.fit / generate /
generate_multi``, ``get_top_k_recall``) loaded through ``oracle/ref_loader.py``.
Only the three third-party pieces that are not in ``/root/reference`` are the
restatements of this package: ``torch.hub.load('facebookresearch/dinov2', ..)``
is redirected to ``oracle.dinov2_ref`` carrying seeded synthetic weights,
``fast_pytorch_kmeans`` -> ``oracle.fpk_kmeans`` and ``faiss`` ->
``oracle.faiss_flat``.  Outputs: small ``.npz`` fixtures under ``tests/golden/``
(inputs are NOT stored: they are regenerated from seeds by ``anyloc_amd.synth``).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from anyloc_amd import synth                     # noqa: E402
from oracle import dinov2_ref, ref_loader        # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def probe_vector(dim, seed=123):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(dim, generator=g)


def hub_redirect(weights_seed):
    """Replacement for ``torch.hub.load`` used while constructing the
    reference's ``DinoV2ExtractFeatures`` (``utilities.py:239-240``)."""
    def _load(repo, name, *a, **k):
        assert repo == "facebookresearch/dinov2"
        return dinov2_ref.build(name, synth.synthetic_state_dict(name, weights_seed))
    return _load


@torch.no_grad()
def golden_config1(ref):
    """BASELINE.json configs[0]: ViT-S/14 layer 9 'value', K=8, 32 synthetic
    224x224 images (24 database + 8 queries), CPU torch."""
    model, layer, facet, K = "dinov2_vits14", 9, "value", 8
    n_db, n_qu, hw = 24, 8, 224
    db_img, qu_img, gt = synth.synthetic_places(n_db, n_qu, hw, hw, seed=42)
    imgs = torch.cat([db_img, qu_img])
    real_hub = torch.hub.load
    torch.hub.load = hub_redirect(0)
    try:
        ext = ref.DinoV2ExtractFeatures(model, layer, facet, device="cpu")
        # the reference drives the extractor one image at a time
        # (scripts/dino_v2_vlad.py:169-183)
        toks = torch.cat([ext(im[None]) for im in imgs])            # [32,256,384]
        variants = {}
        for f in ("query", "key", "token"):
            e2 = ref.DinoV2ExtractFeatures(model, layer, f, device="cpu")
            variants[f] = e2(imgs[:1])[0]
        e3 = ref.DinoV2ExtractFeatures(model, layer, "value", use_cls=True,
                                       norm_descs=False, device="cpu")
        variants["value_cls_raw"] = e3(imgs[:1])[0]
        e4 = ref.DinoV2ExtractFeatures(model, 11, "token", device="cpu")
        variants["token_l11"] = e4(imgs[:1])[0]
    finally:
        torch.hub.load = real_hub
    pv = probe_vector(toks.shape[-1])
    # vocabulary: VLAD.fit on the database tokens (scripts/dino_v2_vlad.py:203-213)
    ref.seed_everything(42)
    init_idx = np.random.RandomState(42).choice(n_db * toks.shape[1], size=[K], replace=False)
    vlad = ref.VLAD(K, None, cache_dir=None)
    vlad.fit(toks[:n_db].reshape(-1, toks.shape[-1]))
    centers = vlad.c_centers.clone()
    labels = torch.stack([vlad.kmeans.predict(t) for t in toks])
    vlads = vlad.generate_multi(toks)                              # [32, 3072]
    svlad = ref.VLAD(K, None, vlad_mode="soft", soft_temp=1.0, cache_dir=None)
    svlad.kmeans, svlad.c_centers, svlad.desc_dim = vlad.kmeans, centers, centers.shape[1]
    soft = torch.stack([svlad.generate(toks[0]), svlad.generate(toks[31])])
    top_k = list(range(1, 21))
    dist, idx, rec = ref.get_top_k_recall(top_k, vlads[:n_db], vlads[n_db:], gt)
    dist_l2, idx_l2, rec_l2 = ref.get_top_k_recall(top_k, vlads[:n_db], vlads[n_db:], gt,
                                                   method="l2")
    out = dict(
        model=model, layer=layer, facet=facet, K=K, n_db=n_db, n_qu=n_qu, hw=hw,
        weights_seed=0, images_seed=42,
        tokens_img0=toks[0].numpy(), tokens_img31=toks[31].numpy(),
        token_proj=(toks @ pv).numpy(),                               # [32,256]
        token_sumsq=(toks.double() ** 2).sum(-1).float().numpy(),
        init_idx=init_idx.astype(np.int64), centers=centers.numpy(),
        kmeans_iters=np.int64(vlad.kmeans.n_iter_),
        labels=labels.numpy().astype(np.int16), vlads=vlads.numpy(),
        soft_vlads=soft.numpy(),
        top_dist=np.asarray(dist), top_idx=np.asarray(idx),
        recalls=np.array([rec[k] for k in top_k], dtype=np.float64),
        top_dist_l2=np.asarray(dist_l2), top_idx_l2=np.asarray(idx_l2),
        recalls_l2=np.array([rec_l2[k] for k in top_k], dtype=np.float64),
    )
    for k, v in variants.items():
        out[f"facet_{k}_proj"] = (v @ pv).numpy()
        out[f"facet_{k}_shape"] = np.array(v.shape)
        out[f"facet_{k}_head"] = v[:4, :16].numpy()
    np.savez_compressed(os.path.join(GOLDEN, "config1_vits14_l9_value_k8.npz"), **out)
    print("config1: recalls", rec, "kmeans iters", vlad.kmeans.n_iter_)


@torch.no_grad()
def golden_vlad_shapes(ref):
    """Reference ``VLAD.generate`` on seeded clustered tokens at the ViT-g/14
    322x322 K=32 shape (configs[1-2]) and the ViT-L/14 518x518 K=64 shape
    (configs[4], single layer) -- inputs regenerated from seeds by the tests."""
    for tag, (n_img, N, D, K, seed) in {
        "c2_n529_d1536_k32": (3, 529, 1536, 32, 7),
        "c5_n1369_d1024_k64": (2, 1369, 1024, 64, 8),
    }.items():
        x = synth.clustered_tokens(n_img, N, D, n_modes=K + 5, seed=seed)
        g = torch.Generator().manual_seed(seed + 100)
        # centres as k-means leaves them: cluster means of unit vectors (norm < 1)
        centers = 0.8 * torch.nn.functional.normalize(torch.randn(K, D, generator=g), dim=1) \
            + 0.02 * torch.randn(K, D, generator=g)
        centers[:K - 5] = synth.clustered_tokens(1, K - 5, D, n_modes=K + 5, seed=seed)[0] * 0.85
        vlad = ref.VLAD(K, D, cache_dir=None)
        vlad.kmeans = type(ref.fpk.KMeans(K, mode="cosine"))(K, mode="cosine")
        vlad.kmeans.centroids = centers
        vlad.c_centers = centers
        vl = vlad.generate_multi(x)
        lab = torch.stack([vlad.kmeans.predict(t) for t in x])
        # raw (not unit-norm) tokens exercise the norm_descs path
        xr = x * (0.5 + torch.rand(n_img, N, 1, generator=g))
        vl_raw = vlad.generate_multi(xr)
        np.savez_compressed(
            os.path.join(GOLDEN, f"vlad_{tag}.npz"),
            n_img=n_img, N=N, D=D, K=K, seed=seed, centers=centers.numpy(),
            scale=(xr[:, :, 0] / x[:, :, 0]).numpy(),
            labels=lab.numpy().astype(np.int16), vlads=vl.numpy(), vlads_raw=vl_raw.numpy())
        print(tag, "used clusters per image:", [len(set(l.tolist())) for l in lab])


@torch.no_grad()
def golden_kmeans(ref):
    """Reference ``VLAD.fit`` (k-means vocabulary) on a seeded mixture."""
    n, D, K = 20000, 64, 16
    x = synth.clustered_tokens(1, n, D, n_modes=K, seed=11, noise=0.6)[0]
    ref.seed_everything(42)
    init_idx = np.random.RandomState(42).choice(n, size=[K], replace=False)
    vlad = ref.VLAD(K, None, cache_dir=None)
    vlad.fit(x)
    np.savez_compressed(os.path.join(GOLDEN, "kmeans_n20000_d64_k16.npz"),
                        n=n, D=D, K=K, seed=11, init_idx=init_idx.astype(np.int64),
                        centers=vlad.c_centers.numpy(),
                        iters=np.int64(vlad.kmeans.n_iter_))
    print("kmeans iters", vlad.kmeans.n_iter_)


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    ref = ref_loader.load_reference_utilities("root")
    golden_config1(ref)
    golden_vlad_shapes(ref)
    golden_kmeans(ref)


if __name__ == "__main__":
    main()
