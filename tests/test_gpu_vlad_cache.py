"""VLAD residual / cache surface on the HIP kernels (GPU box): ``generate_res_vec`` / ``generate_multi_res_vec``
(reference utilities.py:928-1008), ``fit_and_generate`` (:793-817) and the ``cache_dir`` protocol of ``generate``
(:843-852, :864-878: ``<id>_r.pt`` + ``<id>_l.pt`` / ``<id>_s.pt``) against the CPU oracle.  By default the cache is
written the way the reference writes it (dense [N,K,D] ``<id>_r.pt``, readable by the reference's own indexing); the
opt-in compact format (``cache_format = "lazy"``: normalised tokens in ``<id>_t.pt``, the residual tensor is never stored
or rebuilt on a hit) lives under its own file name so a reference-side reader never opens it."""
import os

import numpy as np
import pytest
import torch

from anyloc_amd import synth
from oracle import vlad_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
VLAD_RTOL = 1e-5


def l2rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _fitted(K, D, mode="hard", cache_dir=None, seed=0, **kw):
    import utilities
    g = torch.Generator().manual_seed(seed)
    centers = 0.7 * synth.clustered_tokens(1, K, D, n_modes=K, seed=seed + 1)[0] + 0.01 * torch.randn(K, D, generator=g)
    v = utilities.VLAD(K, D, vlad_mode=mode, cache_dir=cache_dir, **kw)
    v.c_centers = centers
    v.kmeans = utilities.KMeans(K, mode="cosine")
    v.kmeans.centroids = centers
    return v, centers


@pytest.mark.parametrize("K,D,N", [(32, 1536, 529), (8, 384, 256), (5, 64, 77)])
def test_generate_res_vec_vs_reference_expression(K, D, N):
    v, centers = _fitted(K, D)
    g = torch.Generator().manual_seed(N)
    x = synth.clustered_tokens(2, N, D, n_modes=K, seed=3) * (0.5 + torch.rand(2, N, 1, generator=g))
    res = v.generate_res_vec(x[0])                                       # CPU in -> CPU out
    ref = torch.nn.functional.normalize(x[0])[:, None, :] - centers[None]            # utilities.py:959-962
    assert res.device.type == "cpu" and res.shape == (N, K, D)
    assert float((res - ref).abs().max()) < 2e-7
    res_dev = v.generate_res_vec(x[1].to(DEV))                           # device in -> device out
    assert res_dev.is_cuda
    assert float((res_dev.cpu() - (torch.nn.functional.normalize(x[1])[:, None, :] - centers[None])).abs().max()) < 2e-7
    multi = v.generate_multi_res_vec(x)                                  # utilities.py:974-1008: stacked [n_img,N,K,D]
    assert tuple(multi.shape) == (2, N, K, D) and torch.equal(multi[0], res)
    v2, _ = _fitted(K, D, norm_descs=False)
    assert float((v2.generate_res_vec(x[0]) - (x[0][:, None, :] - centers[None])).abs().max()) < 2e-7
    # summing the own-cluster residuals reproduces the VLAD (the reference's definition, utilities.py:854-861)
    vl, lab = vlad_ref.vlad_hard(x[0], centers)
    assert l2rel(v.generate(x[0]), vl) < VLAD_RTOL


@pytest.mark.parametrize("fmt", ["reference", "lazy"])
@pytest.mark.parametrize("mode", ["hard", "soft"])
def test_cache_dir_round_trip(tmp_path, mode, fmt):
    K, D, N = (32, 1536, 529) if mode == "hard" else (8, 384, 200)
    cache = str(tmp_path / "cache")
    v, centers = _fitted(K, D, mode, cache_dir=None)
    x = synth.clustered_tokens(3, N, D, n_modes=K, seed=9)
    direct = v.generate_multi(x)
    vc, _ = _fitted(K, D, mode, cache_dir=cache)
    assert vc.cache_format == "reference"                                # the default is the reference's own file format
    vc.cache_format = fmt
    torch.save(centers, f"{cache}/c_centers.pt")                         # what VLAD.fit leaves behind (utilities.py:788-791)
    assert vc.can_use_cache_vlad()
    ids = [f"db/img{i}" for i in range(3)]
    assert not vc.can_use_cache_ids(ids)
    first = vc.generate_multi(x, ids)                                    # miss: computes and writes _r + _l / _s
    assert l2rel(first, direct) < 1e-6
    assert vc.can_use_cache_ids(ids)
    if fmt == "lazy":
        # the compact file: normalised tokens, K times smaller than the reference's tensor, and NOT named <id>_r.pt
        assert not os.path.exists(f"{cache}/{ids[0]}_r.pt")
        r_obj = torch.load(f"{cache}/{ids[0]}_t.pt")
        assert isinstance(r_obj, dict) and tuple(r_obj["tokens"].shape) == (N, D)
        assert os.path.getsize(f"{cache}/{ids[0]}_t.pt") < 1.1 * N * D * 4 + 4096
    else:
        # what a reference-side reader does with the file (utilities.py:843-847): torch.load + residuals[labels == k, k]
        r_obj = torch.load(f"{cache}/{ids[0]}_r.pt")
        assert torch.is_tensor(r_obj) and tuple(r_obj.shape) == (N, K, D) and not os.path.exists(f"{cache}/{ids[0]}_t.pt")
        assert float((r_obj - (torch.nn.functional.normalize(x[0])[:, None, :] - centers[None])).abs().max()) < 2e-7
    side = torch.load(f"{cache}/{ids[0]}_{'l' if mode == 'hard' else 's'}.pt")
    if mode == "hard":
        assert torch.equal(side, vlad_ref.hard_labels(x[0], centers))
    else:
        assert float((side - vlad_ref.vlad_soft(x[0], centers, 1.0)[1]).abs().max()) < 1e-6
    hit = vc.generate_multi([None, None, None], ids)                     # hit: no descriptors needed at all
    for i in range(3):
        ref = (vlad_ref.vlad_hard if mode == "hard" else lambda t, c: vlad_ref.vlad_soft(t, c, 1.0))(x[i], centers)[0]
        assert l2rel(hit[i], ref) < VLAD_RTOL
        assert l2rel(hit[i], direct[i]) < 2e-6
    # generate_res_vec on a hit: the stored tensor, or rebuilt from the compact file
    res = vc.generate_res_vec(None, ids[1])
    assert float((res - (torch.nn.functional.normalize(x[1])[:, None, :] - centers[None])).abs().max()) < 2e-7


def test_dense_cache_written_like_the_reference_is_honoured(tmp_path):
    """<id>_r.pt as the reference writes it ([N,K,D] tensor, utilities.py:963-970) + <id>_l.pt: the hit reads only the
    own-cluster slice of every token; cache_format='reference' makes this class write that format itself."""
    K, D, N = 16, 384, 300
    cache = str(tmp_path / "cache")
    v, centers = _fitted(K, D, cache_dir=cache)
    torch.save(centers, f"{cache}/c_centers.pt")
    x = synth.clustered_tokens(1, N, D, n_modes=K, seed=4)[0]
    torch.save(torch.nn.functional.normalize(x)[:, None, :] - centers[None], f"{cache}/q0_r.pt")
    torch.save(vlad_ref.hard_labels(x, centers), f"{cache}/q0_l.pt")
    out = v.generate(None, "q0")
    assert l2rel(out, vlad_ref.vlad_hard(x, centers)[0]) < VLAD_RTOL
    assert torch.equal(v.generate_res_vec(None, "q0"), torch.load(f"{cache}/q0_r.pt"))
    assert v.cache_format == "reference"
    v.generate(x, "q1")
    dense = torch.load(f"{cache}/q1_r.pt")
    assert torch.is_tensor(dense) and tuple(dense.shape) == (N, K, D)
    assert float((dense - torch.load(f"{cache}/q0_r.pt")).abs().max()) < 2e-7
    # soft mode on a dense file: the reference's all-cluster sum
    vs, _ = _fitted(K, D, "soft", cache_dir=cache)
    w = vlad_ref.vlad_soft(x, centers, 1.0)[1]
    torch.save(w, f"{cache}/q0_s.pt")
    assert l2rel(vs.generate(None, "q0"), vlad_ref.vlad_soft(x, centers, 1.0)[0]) < VLAD_RTOL


def test_fit_and_generate_and_assigned_kernel(tmp_path):
    import utilities
    from anyloc_amd import ops
    K, D, N = 8, 384, 256
    x = synth.clustered_tokens(6, N, D, n_modes=K, seed=12)
    np.random.seed(3)
    v = utilities.VLAD(K, None, cache_dir=None)
    out = v.fit_and_generate(x)                                           # utilities.py:793-817
    assert tuple(out.shape) == (6, K * D) and v.desc_dim == D
    # north-star bar on every image; an image may exceed it only through a PROVEN tie: a cluster id that differs from the
    # oracle's at an oracle top-2 cosine gap below fp32 resolution (then the oracle VLAD under the kernel's ids must match)
    _, lab_all = ops.vlad(x.to(DEV), v.c_centers.to(DEV), return_labels=True)
    lab_all = lab_all.cpu().reshape(6, N)
    for i in range(6):
        lab_ref = vlad_ref.hard_labels(x[i], v.c_centers)
        if torch.equal(lab_all[i], lab_ref):
            assert l2rel(out[i], vlad_ref.vlad_hard(x[i], v.c_centers)[0]) < VLAD_RTOL
        else:
            sc = vlad_ref.fpk_cosine_scores(x[i][lab_all[i] != lab_ref], v.c_centers).topk(2, dim=1)[0]
            assert float((sc[:, 0] - sc[:, 1]).max()) < 1e-6
            assert l2rel(out[i], vlad_ref.vlad_hard(x[i], v.c_centers, labels=lab_all[i])[0]) < VLAD_RTOL
    # ops.vlad_assigned == ops.vlad when fed the labels ops.vlad chose itself
    c = v.c_centers.to(DEV)
    full, lab = ops.vlad(x[:1].to(DEV), c, return_labels=True)
    again = ops.vlad_assigned(x[0].to(DEV), c, labels=lab)
    assert l2rel(again, full[0]) < 1e-6
    w = ops.vlad_soft_weights(x[0].to(DEV), c, 2.0)
    soft_full = ops.vlad(x[:1].to(DEV), c, mode="soft", soft_temp=2.0)
    assert l2rel(ops.vlad_assigned(x[0].to(DEV), c, soft=w), soft_full[0]) < 1e-6
    with pytest.raises(ValueError):
        ops.vlad_assigned(x[0].to(DEV), c, labels=torch.full((N,), K, dtype=torch.int64))
