"""The oracle restatements (oracle/*.py) reproduce the golden vectors recorded from the
REFERENCE's own code (oracle/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from anyloc_amd import synth
from oracle import dinov2_ref, vlad_ref
from oracle.make_golden import probe_vector

torch.set_num_threads(max(1, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def g1(golden_dir):
    return np.load(os.path.join(golden_dir, "config1_vits14_l9_value_k8.npz"))


@pytest.fixture(scope="module")
def c1_tokens(g1):
    """Oracle ViT-S/14 L9 value tokens of the 32 synthetic config-1 images."""
    sd = synth.synthetic_state_dict(str(g1["model"]), int(g1["weights_seed"]))
    model = dinov2_ref.build(str(g1["model"]), sd)
    db, qu, gt = synth.synthetic_places(int(g1["n_db"]), int(g1["n_qu"]), int(g1["hw"]), int(g1["hw"]),
                                        seed=int(g1["images_seed"]))
    imgs = torch.cat([db, qu])
    toks = torch.cat([dinov2_ref.extract_facet(model, im[None], int(g1["layer"]), str(g1["facet"]))
                      for im in imgs])
    return toks, gt, model, imgs


def test_dinov2_restatement_matches_reference_hook_path(g1, c1_tokens):
    toks = c1_tokens[0]
    # bitwise: same torch CPU kernels, same order of operations as the reference's __call__
    assert np.array_equal(toks[0].numpy(), g1["tokens_img0"])
    assert np.array_equal(toks[31].numpy(), g1["tokens_img31"])
    pv = probe_vector(toks.shape[-1])
    assert np.array_equal((toks @ pv).numpy(), g1["token_proj"])


def test_facet_variants(g1, c1_tokens):
    _, _, model, imgs = c1_tokens
    pv = probe_vector(384)
    for name, kw in {"query": dict(layer=9, facet="query"), "key": dict(layer=9, facet="key"),
                     "token": dict(layer=9, facet="token"),
                     "value_cls_raw": dict(layer=9, facet="value", use_cls=True, norm_descs=False),
                     "token_l11": dict(layer=11, facet="token")}.items():
        out = dinov2_ref.extract_facet(model, imgs[:1], **kw)[0]
        assert tuple(out.shape) == tuple(g1[f"facet_{name}_shape"])
        assert np.array_equal((out @ pv).numpy(), g1[f"facet_{name}_proj"])


def test_early_exit_equals_full_forward(c1_tokens):
    """Stopping after the hooked block gives the hooked tensor of the full forward."""
    toks, _, model, imgs = c1_tokens
    grabbed = {}
    h = model.blocks[9].attn.qkv.register_forward_hook(lambda m, i, o: grabbed.__setitem__("o", o))
    with torch.no_grad():
        model(imgs[:1], n_blocks=10)
    h.remove()
    v = torch.nn.functional.normalize(grabbed["o"][:, 1:, 768:], dim=-1)
    assert torch.equal(v[0], toks[0])


def test_kmeans_vocabulary(g1, c1_tokens):
    toks = c1_tokens[0]
    n_db = int(g1["n_db"])
    centers, iters = vlad_ref.kmeans_fit(toks[:n_db].reshape(-1, toks.shape[-1]), int(g1["K"]),
                                         init_idx=g1["init_idx"])
    assert iters == int(g1["kmeans_iters"])
    assert np.array_equal(centers.numpy(), g1["centers"])


def test_vlad_hard_soft_and_recall(g1, c1_tokens):
    toks, gt = c1_tokens[0], c1_tokens[1]
    centers = torch.from_numpy(g1["centers"])
    vl, lab = zip(*[vlad_ref.vlad_hard(t, centers) for t in toks])
    vl = torch.stack(vl)
    assert np.array_equal(torch.stack(lab).numpy(), g1["labels"].astype(np.int64))
    assert np.array_equal(vl.numpy(), g1["vlads"])
    soft = torch.stack([vlad_ref.vlad_soft(toks[0], centers)[0], vlad_ref.vlad_soft(toks[31], centers)[0]])
    np.testing.assert_allclose(soft.numpy(), g1["soft_vlads"], rtol=0, atol=1e-7)
    n_db = int(g1["n_db"])
    top_k = list(range(1, 21))
    for method, sfx in (("cosine", ""), ("l2", "_l2")):
        d, i, r = vlad_ref.top_k_recall(top_k, vl[:n_db], vl[n_db:], gt, method=method)
        assert np.array_equal(i.numpy(), g1["top_idx" + sfx])
        np.testing.assert_allclose(d.numpy(), g1["top_dist" + sfx], rtol=0, atol=1e-6)
        assert [r[k] for k in top_k] == list(g1["recalls" + sfx])


@pytest.mark.parametrize("tag", ["c2_n529_d1536_k32", "c5_n1369_d1024_k64"])
def test_vlad_shapes(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"vlad_{tag}.npz"))
    x = synth.clustered_tokens(int(g["n_img"]), int(g["N"]), int(g["D"]), n_modes=int(g["K"]) + 5,
                               seed=int(g["seed"]))
    centers = torch.from_numpy(g["centers"])
    for i in range(x.shape[0]):
        v, l = vlad_ref.vlad_hard(x[i], centers)
        assert np.array_equal(l.numpy(), g["labels"][i].astype(np.int64))
        assert np.array_equal(v.numpy(), g["vlads"][i])
        vr, _ = vlad_ref.vlad_hard(x[i] * torch.from_numpy(g["scale"][i])[:, None], centers)
        np.testing.assert_allclose(vr.numpy(), g["vlads_raw"][i], rtol=0, atol=2e-7)


def test_kmeans_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "kmeans_n20000_d64_k16.npz"))
    x = synth.clustered_tokens(1, int(g["n"]), int(g["D"]), n_modes=int(g["K"]), seed=int(g["seed"]),
                               noise=0.6)[0]
    centers, iters = vlad_ref.kmeans_fit(x, int(g["K"]), init_idx=g["init_idx"])
    assert iters == int(g["iters"])
    assert np.array_equal(centers.numpy(), g["centers"])
