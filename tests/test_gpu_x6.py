"""The split GEMM paths through the C ABI:
  * x6 -- exact three-way bf16 splits, six bf16 MFMA products (anyloc_split_x3 / anyloc_gemm_nt_x6 / anyloc_vit_attach_x3),
  * h3 -- row-scaled two-term fp16 splits, three fp16 MFMA products (anyloc_split_h2 / anyloc_gemm_nt_h3 /
          anyloc_vit_attach_h2; the default of the Python surface).
The splits are exact (x6) / 22-bit relative to the row maximum (h3), both GEMMs are as accurate as an fp32 GEMM
(measured against float64), and the ViT forward gives the same tokens in all three GEMM modes and matches the golden
vectors in each; fused / alternative kernel variants are bit-identical to the ones they replace."""
import os

import numpy as np
import pytest
import torch

from anyloc_amd import synth, weights
from anyloc_amd import ops as ops_mod

pytestmark = pytest.mark.gpu
DEV = "cuda"


def planes_from_image(img3, rows, K):
    """Inverse of the x3 layout (csrc/gemm_x6.hip): [kb][plane][row][16] bf16, 16-byte halves of a row swapped
    when (row >> 3) & 1  ->  three float32 matrices [rows, 16*K16]."""
    k16 = (K + 15) // 16
    t = img3.view(torch.bfloat16).reshape(k16, 3, rows, 2, 8).clone()
    odd = ((torch.arange(rows, device=img3.device) >> 3) & 1).bool()
    t[:, :, odd] = t[:, :, odd].flip(3)
    return t.permute(1, 2, 0, 3, 4).reshape(3, rows, k16 * 16).float()


@pytest.mark.parametrize("shape", [(128, 16), (300, 100), (1000, 384), (77, 1536), (1, 5)])
def test_split_is_exact(shape):
    from anyloc_amd import ops
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(*shape, generator=g, device=DEV) * torch.exp(4 * torch.randn(shape[0], 1, generator=g, device=DEV))
    x[0, 0] = 0.0
    x[-1, -1] = 1.0 + 2.0 ** -23                                          # needs all 24 bits
    p = planes_from_image(ops.split_x3(x), *shape)
    assert p.shape[2] % 16 == 0
    back = (p[0].double() + p[1].double() + p[2].double())[:, :shape[1]]
    assert torch.equal(back, x.double())                                   # x = x1 + x2 + x3 exactly
    assert float(p[:, :, shape[1]:].abs().max() if p.shape[2] > shape[1] else 0.0) == 0.0   # k padding is zero
    assert float((p[1].abs() > p[0].abs() * 2.0 ** -7 + 1e-38).float().max()) == 0.0      # |x2| <= ulp_bf16(x1)
    # leading plane = round-to-nearest-even bf16 of x
    assert torch.equal(p[0][:, :shape[1]], x.to(torch.bfloat16).float())


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 200, 48), (130, 515, 100), (1000, 384, 384), (2051, 1536, 1536),
                                   (64, 4608, 1536), (5, 7, 3), (4096, 4096, 4096)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_gemm_x6_as_accurate_as_fp32(M, N, K, with_bias):
    from anyloc_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device=DEV) * (0.25 + torch.rand(M, 1, generator=g, device=DEV))
    w = torch.randn(N, K, generator=g, device=DEV) * 0.05
    bias = torch.randn(N, generator=g, device=DEV) if with_bias else None
    c = ops.gemm_nt_x6(ops.split_x3(a), ops.split_x3(w), M, N, K, bias)
    ref = a.double() @ w.double().t() + (bias.double() if with_bias else 0.0)
    mag = a.double().abs() @ w.double().abs().t() + (bias.double().abs() if with_bias else 0.0)
    err = float(((c.double() - ref).abs() / mag).max())
    # the dropped plane products are < 2^-23 of |a||b| per term; fp32 accumulation adds ~sqrt(K) * 2^-24
    assert err < 6e-7, err
    if K % 4 == 0:
        c32 = ops.gemm_nt(a, w, bias)
        e32 = float(((c32.double() - ref).abs() / mag).max())
        assert err < 1.5 * e32 + 1e-7, (err, e32)                          # no worse than the fp32-MFMA kernel


def h2_planes_from_image(img2, rows, K):
    """Inverse of the h2 layout (csrc/gemm_h3.hip): [kb][plane][row][16] fp16 with the x3 half swap -> [2, rows, K]."""
    k16 = K // 16
    t = img2.view(torch.float16).reshape(k16, 2, rows, 2, 8).clone()
    odd = ((torch.arange(rows, device=img2.device) >> 3) & 1).bool()
    t[:, :, odd] = t[:, :, odd].flip(3)
    return t.permute(1, 2, 0, 3, 4).reshape(2, rows, K).float()


@pytest.mark.parametrize("shape", [(128, 16), (300, 96), (1000, 384), (77, 1536), (33, 4096)])
def test_split_h2_row_scaling_and_precision(shape):
    from anyloc_amd import ops
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(*shape, generator=g, device=DEV) * torch.exp(3 * torch.randn(shape[0], 1, generator=g, device=DEV))
    x[:, ::7] *= 40.0                                                       # heavy-tailed columns in every row
    x[5] = 0.0                                                              # an all-zero row
    img, inv = ops.split_h2(x)
    p = h2_planes_from_image(img, *shape)
    amax = x.abs().amax(dim=1)
    scaled = amax / inv                                                     # amax * 2^e
    ok = amax > 0
    assert bool(((scaled[ok] >= 2.0 ** 14) & (scaled[ok] < 2.0 ** 15)).all())
    assert bool((torch.log2(inv) == torch.log2(inv).round()).all())         # powers of two
    assert float(p.abs().max()) < 65504.0
    back = (p[0].double() + p[1].double()) * inv.double()[:, None]
    err = (back - x.double()).abs().amax(dim=1)
    assert float((err / amax.clamp_min(1e-30)).max()) < 2.0 ** -22         # 22 bits relative to the row maximum
    assert float(back[5].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 200, 48), (130, 515, 112), (2051, 1536, 1536), (64, 4608, 1536),
                                   (4096, 1536, 4096)])
def test_gemm_h3_as_accurate_as_fp32(M, N, K):
    from anyloc_amd import ops
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g, device=DEV) * (0.25 + torch.rand(M, 1, generator=g, device=DEV))
    a[:, ::53] *= 30.0
    w = torch.randn(N, K, generator=g, device=DEV) * 0.05
    bias = torch.randn(N, generator=g, device=DEV)
    c = ops.gemm_nt_h3(ops.split_h2(a), ops.split_h2(w), M, N, K, bias)
    ref = a.double() @ w.double().t() + bias.double()
    mag = a.double().abs() @ w.double().abs().t() + bias.double().abs()
    err = float(((c.double() - ref).abs() / mag).max())
    e32 = float(((ops.gemm_nt(a, w, bias).double() - ref).abs() / mag).max())
    assert err < 1.5 * e32 + 1e-7, (err, e32)
    assert torch.equal(c, ops.gemm_nt_h3(ops.split_h2(a), ops.split_h2(w), M, N, K, bias))


@pytest.mark.parametrize("name,layer,depth,hw", [("dinov2_vitg14", 1, 2, (322, 322)), ("dinov2_vits14", 5, 6, (224, 308))])
def test_h3_epilogue_and_quantiser_variants_are_bit_identical(monkeypatch, name, layer, depth, hw):
    """The LDS-transposed 16-byte LayerScale-residual epilogue vs the dword one (option h3_epi_lds = 0) does the same
    arithmetic in the same order: identical bits."""
    import utilities
    monkeypatch.setenv("ANYLOC_GEMM", "h3")
    ops_mod.set_option("x6_min_rows", 0)
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, depth=depth))
    try:
        imgs = torch.cat(synth.synthetic_places(4, 1, hw[0], hw[1], seed=13)[:2]).to(DEV)
        base = utilities.DinoV2ExtractFeatures(name, layer, "token", use_cls=True, norm_descs=False, device=DEV)(imgs)
        with ops_mod.options(h3_epi_lds=0):
            a = utilities.DinoV2ExtractFeatures(name, layer, "token", use_cls=True, norm_descs=False, device=DEV)(imgs)
        assert torch.isfinite(base).all()
        assert torch.equal(base, a)
    finally:
        weights.unregister_state_dict()


def test_gemm_x6_deterministic_and_tail_rows_untouched():
    from anyloc_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    a, w = torch.randn(333, 200, generator=g, device=DEV), torch.randn(77, 200, generator=g, device=DEV)
    a3, w3 = ops.split_x3(a), ops.split_x3(w)
    c1, c2 = ops.gemm_nt_x6(a3, w3, 333, 77, 200), ops.gemm_nt_x6(a3, w3, 333, 77, 200)
    assert torch.equal(c1, c2)
    # a row sub-range of A's image is addressable: first 128 rows only
    c_head = ops.gemm_nt_x6(ops.split_x3(a[:128].contiguous()), w3, 128, 77, 200)
    assert torch.equal(c_head, c1[:128])


@pytest.mark.parametrize("name,layer,depth", [("dinov2_vits14", 9, None), ("dinov2_vitg14", 1, 2)])
def test_vit_tokens_agree_between_gemm_modes(monkeypatch, name, layer, depth):
    import utilities
    ops_mod.set_option("x6_min_rows", 0)        # small test batches: force the split-bf16 kernels
    sd = synth.synthetic_state_dict(name, 0, depth=depth)
    weights.register_state_dict(name, sd)
    try:
        imgs = torch.cat(synth.synthetic_places(3, 1, 224, 224, seed=5)[:2]).to(DEV)
        out = {}
        for mode in ("x6", "h3", "f32"):
            monkeypatch.setenv("ANYLOC_GEMM", mode)
            ext = utilities.DinoV2ExtractFeatures(name, layer, "value", device=DEV)
            assert ext.dino_model.gemm == mode
            out[mode] = ext(imgs)
            out[mode + "_tok"] = utilities.DinoV2ExtractFeatures(name, layer, "token", use_cls=True, device=DEV)(imgs)
        for mode in ("x6", "h3"):
            assert float((out[mode] - out["f32"]).abs().max()) < 2e-6, mode          # unit-norm rows
            assert float((out[mode + "_tok"] - out["f32_tok"]).abs().max()) < 2e-6, mode
        assert not torch.equal(out["h3"], out["x6"])                       # really different kernels
    finally:
        weights.unregister_state_dict()


@pytest.mark.parametrize("name,layer,depth,hw", [("dinov2_vits14", 9, None, (224, 224)), ("dinov2_vitg14", 1, 2, (126, 154)),
                                                 ("dinov2_vitl14", 2, 3, (70, 98))])
def test_fused_plane_producers_equal_split_passes(monkeypatch, name, layer, depth, hw):
    """LayerNorm / attention / GELU-SwiGLU epilogue writing the plane image directly vs fp32 activations + a
    split pass (option x6_fuse = 0).  The split is exact, so the only difference is LayerNorm's reduction order
    (per-wave rows vs per-block rows): a few ulp on unit-norm tokens."""
    import utilities
    monkeypatch.setenv("ANYLOC_GEMM", "x6")
    ops_mod.set_option("x6_min_rows", 0)
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, depth=depth))
    try:
        imgs = torch.cat(synth.synthetic_places(4, 1, hw[0], hw[1], seed=9)[:2]).to(DEV)
        res = {}
        for fuse in ("1", "0"):
            ops_mod.set_option("x6_fuse", int(fuse))
            res[fuse] = (utilities.DinoV2ExtractFeatures(name, layer, "value", device=DEV)(imgs),
                         utilities.DinoV2ExtractFeatures(name, layer, "token", use_cls=True, norm_descs=False,
                                                         device=DEV)(imgs))
        assert float((res["1"][0] - res["0"][0]).abs().max()) < 1e-6
        scale = float(res["0"][1].abs().max())
        assert float((res["1"][1] - res["0"][1]).abs().max()) < 2e-6 * scale
    finally:
        weights.unregister_state_dict()


def test_golden_tokens_in_fp32_mfma_mode(monkeypatch, golden_dir):
    """The exact-fp32 MFMA path stays a supported mode (ANYLOC_GEMM=f32): same golden vector, same tolerance."""
    import utilities
    ops_mod.set_option("x6_min_rows", 0)
    g1 = np.load(os.path.join(golden_dir, "config1_vits14_l9_value_k8.npz"))
    name = str(g1["model"])
    weights.register_state_dict(name, synth.synthetic_state_dict(name, int(g1["weights_seed"])))
    try:
        db, qu, _ = synth.synthetic_places(int(g1["n_db"]), int(g1["n_qu"]), int(g1["hw"]), int(g1["hw"]),
                                           seed=int(g1["images_seed"]))
        for mode in ("f32", "x6", "h3"):
            monkeypatch.setenv("ANYLOC_GEMM", mode)
            ext = utilities.DinoV2ExtractFeatures(name, 9, "value", device=DEV)
            one = ext(db[:1].to(DEV))
            assert float((one[0].cpu() - torch.from_numpy(g1["tokens_img0"])).abs().max()) < 2e-5, mode
    finally:
        weights.unregister_state_dict()


def test_small_batches_take_the_fp32_mfma_kernels(monkeypatch):
    """Below option x6_min_rows token rows (default 1600, i.e. B <= 3 at 322x322) a split-bf16 forward runs the
    exact-fp32 kernels (more, smaller tiles): bit-identical to ANYLOC_GEMM=f32, and different bits from the forced
    split-bf16 run of the same batch."""
    import utilities
    name = "dinov2_vits14"
    weights.register_state_dict(name, synth.synthetic_state_dict(name, 0, depth=4))
    try:
        img = synth.synthetic_places(1, 0, 224, 224, seed=3)[0].to(DEV)            # 257 rows
        monkeypatch.setenv("ANYLOC_GEMM", "x6")
        auto = utilities.DinoV2ExtractFeatures(name, 3, "value", device=DEV)(img)
        ops_mod.set_option("x6_min_rows", 0)
        forced = utilities.DinoV2ExtractFeatures(name, 3, "value", device=DEV)(img)
        monkeypatch.setenv("ANYLOC_GEMM", "f32")
        exact = utilities.DinoV2ExtractFeatures(name, 3, "value", device=DEV)(img)
        assert torch.equal(auto, exact)
        assert not torch.equal(forced, exact) and float((forced - exact).abs().max()) < 2e-6
    finally:
        weights.unregister_state_dict()


def test_large_batches_are_chunked_below_the_plane_image_limit(monkeypatch):
    """One operand's plane image must stay inside 2 GiB of buffer addressing: the extractor splits the batch
    (``max_rows``); chunked and unchunked results are identical."""
    from anyloc_amd import extractor
    ops_mod.set_option("x6_min_rows", 0)
    name = "dinov2_vits14"
    model = extractor.HipDinoV2(name, synth.synthetic_state_dict(name, 0, depth=3), torch.device(DEV), gemm="x6")
    imgs = torch.cat(synth.synthetic_places(6, 1, 112, 140, seed=2)[:2]).to(DEV)       # 7 images x 81 tokens
    whole = model.forward_taps(imgs, [(2, "value")])
    model.max_rows = 2 * 81 + 5                                                          # -> chunks of 2 images
    parts = model.forward_taps(imgs, [(2, "value")])
    assert torch.equal(whole, parts)


def test_vit_forward_without_attached_planes_is_an_error():
    from anyloc_amd import _lib, extractor, ops
    name = "dinov2_vits14"
    model = extractor.HipDinoV2(name, synth.synthetic_state_dict(name, 0, depth=2), torch.device(DEV), gemm="f32")
    model.gemm = "x6"                                   # ask for the split path without plane images attached
    with pytest.raises(_lib.AnylocHipError, match="attach_x3"):
        model.forward_taps(torch.zeros(1, 3, 28, 28, device=DEV), [(1, "value")])
