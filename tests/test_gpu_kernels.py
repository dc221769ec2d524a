"""Unit parity of the HIP building blocks against fp64 / torch references (GPU box only).
Called through the C ABI (anyloc_amd.ops -> ctypes -> libanyloc_hip.so)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from anyloc_amd import _lib
    _lib.load()
    return torch.device("cuda")


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("M,N,K", [(300, 256, 64), (129, 384, 588), (1000, 32, 96), (64, 17, 40),
                                   (257, 130, 36), (1, 128, 4), (4240, 1536, 1536), (530, 8192, 1536)])
def test_gemm_nt(dev, M, N, K):
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)          # asymmetric operands: a transposed C would fail
    bias = torch.randn(N, generator=g)
    ref = a.double() @ w.double().T + bias.double()
    out = ops.gemm_nt(a.to(dev), w.to(dev), bias.to(dev)).cpu()
    assert out.shape == (M, N)
    # k-ordered fp32 fma chain vs fp64: random-walk rounding error ~ sqrt(K) * 6e-8 * |partial sums|;
    # the bound is ~15 sigma of that at K = 1536 over millions of outputs (a layout bug is O(1))
    bound = 1.5e-6 * (a.abs().double() @ w.abs().double().T) + 1e-6
    assert bool(((out.double() - ref).abs() <= bound).all()), _rel(out, ref)
    out2 = ops.gemm_nt(a.to(dev), w.to(dev)).cpu()
    assert bool(((out2.double() - (ref - bias.double())).abs() <= bound).all())


def test_gemm_identity_is_exact(dev):
    """A = I picks rows of W exactly (catches operand/accumulator layout swaps)."""
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(1)
    w = torch.randn(160, 192, generator=g)
    eye = torch.eye(192)
    out = ops.gemm_nt(eye.to(dev), w.to(dev)).cpu()     # [192,160] = W^T
    assert torch.equal(out, w.T.contiguous())


@pytest.mark.parametrize("rows,dim", [(7, 384), (529, 1536), (3, 49152), (5, 10), (2, 6)])
def test_l2norm_rows(dev, rows, dim):
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(rows + dim)
    x = torch.randn(rows, dim, generator=g) * 3
    x[0] = 0                                            # zero row stays zero (eps clamp)
    out = ops.l2norm_rows(x.to(dev)).cpu()
    ref = torch.nn.functional.normalize(x.double(), dim=-1)
    assert _rel(out, ref) < 5e-7
    assert torch.equal(out[0], torch.zeros(dim))


@pytest.mark.parametrize("rows,dim", [(11, 384), (530, 1536), (9, 1024)])
def test_layernorm(dev, rows, dim):
    from anyloc_amd import ops
    g = torch.Generator().manual_seed(rows * dim)
    x = torch.randn(rows, dim, generator=g) * 2 + 0.5
    w, b = torch.randn(dim, generator=g), torch.randn(dim, generator=g)
    out = ops.layernorm(x.to(dev), w.to(dev), b.to(dev), 1e-6).cpu()
    ref = torch.nn.functional.layer_norm(x.double(), (dim,), w.double(), b.double(), 1e-6)
    assert float((out.double() - ref).abs().max()) < 5e-6


@pytest.mark.parametrize("kernel", ["fp32-mfma", "split-bf16"])
@pytest.mark.parametrize("B,T,heads", [(2, 257, 6), (1, 530, 24), (3, 33, 2), (1, 128, 1), (2, 1370, 2)])
def test_attention(dev, B, T, heads, kernel, monkeypatch):
    """Both attention kernels behind anyloc_attention against float64: the exact-fp32 MFMA one and the
    six-product split-bf16 one the x6 forward uses (option attn_x6 selects it for anyloc_attention)."""
    from anyloc_amd import ops
    ops.set_option("attn_x6", 1 if kernel == "split-bf16" else 0)
    D = heads * 64
    g = torch.Generator().manual_seed(B * T + heads)
    qkv = torch.randn(B, T, 3 * D, generator=g) * 1.5
    qkv[0, 3, :D] *= 6.0      # a spiky query/key pair: forces the running-max rescale path
    qkv[0, T - 2, D:2 * D] *= 6.0
    out = ops.attention(qkv.to(dev), heads).cpu()
    q, k, v = qkv.double().reshape(B, T, 3, heads, 64).permute(2, 0, 3, 1, 4)
    a = torch.softmax((q * 0.125) @ k.transpose(-2, -1), dim=-1)
    ref = (a @ v).transpose(1, 2).reshape(B, T, D)
    assert float((out.double() - ref).abs().max()) < 2e-5, float((out.double() - ref).abs().max())
    assert _rel(out, ref) < 5e-6


@pytest.mark.parametrize("qg,ks", [(1, 1), (2, 1), (1, 2)], ids=["4x32q", "2x64q", "2x32q-x-2keys"])
@pytest.mark.parametrize("B,T,heads", [(2, 257, 6), (1, 530, 24), (3, 33, 2), (1, 128, 1), (2, 1370, 2), (5, 530, 4), (1, 20, 3)])
def test_attention_h3(dev, B, T, heads, qg, ks):
    """The attention kernel of the two-term fp16 forward (anyloc_attention_h3: per-(head, 32-row group) scaled fp16
    tiles, three matrix-core products per contraction, output written as the h2 image of the projection GEMM)
    against float64.  Tokens of very different magnitude share 32-row groups and images share groups (T % 32 != 0)."""
    from anyloc_amd import ops
    ops.set_option("attn_h3_qg", qg)                      # 2 = the two-waves-of-64-queries workgroup shape (A/B variant)
    ops.set_option("attn_h3_ks", ks)                      # 2 = two query waves x two key waves (the shape of few-image calls)
    D = heads * 64
    g = torch.Generator().manual_seed(B * T + heads)
    qkv = torch.randn(B, T, 3 * D, generator=g) * 1.5
    qkv[0, 3, :D] *= 6.0
    qkv[0, T - 2, D:2 * D] *= 6.0
    qkv[:, ::7] *= 0.05                                   # small-magnitude tokens next to ordinary ones
    qkv[:, 5, 2 * D:] *= 40.0                             # one value row far above the others (sets the image scale)
    if T > 64:
        qkv[0, 32:64, 2 * D:] = 0.0                       # an all-zero V tile (global rows 32..63): must not set the scale
    img, inv = ops.attention_h3(qkv.to(dev), heads)
    out = ops.h2_image_to_f32(img, inv, B * T, D).reshape(B, T, D).cpu()
    q, k, v = qkv.double().reshape(B, T, 3, heads, 64).permute(2, 0, 3, 1, 4)
    a = torch.softmax((q * 0.125) @ k.transpose(-2, -1), dim=-1)
    ref = (a @ v).transpose(1, 2).reshape(B, T, D)
    # every row of an image carries 22 bits relative to the image's largest |v|: error <= 2^-22 of that bound
    vmax = qkv[:, :, 2 * D:].abs().amax(dim=(1, 2)).double()
    err = (out - ref).abs().amax(dim=(1, 2))
    assert bool((err <= 3e-6 * vmax + 2e-6).all()), (err, vmax)
    assert _rel(out, ref) < 5e-6
    assert bool(torch.isfinite(out).all())
    inv_c = inv.cpu().reshape(B, T)
    assert bool((inv_c == inv_c[:, :1]).all())            # one power-of-two scale per image
    assert bool((torch.log2(inv_c) == torch.log2(inv_c).round()).all())


@pytest.mark.parametrize("H,W", [(224, 224), (333, 481), (126, 155)])
def test_preprocess_u8_matches_totensor_normalize_centercrop(dev, H, W):
    """uint8 HWC -> float CHW ingest == ToTensor + Normalize + CenterCrop(h//14*14, w//14*14), bit-exact."""
    from anyloc_amd import preprocess, synth
    g = torch.Generator().manual_seed(H + W)
    u8 = torch.randint(0, 256, (3, H, W, 3), generator=g, dtype=torch.uint8)
    out = preprocess.images_to_input(u8).cpu()
    x = u8.permute(0, 3, 1, 2).float().div(255)
    mean = torch.tensor(synth.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(synth.IMAGENET_STD).view(1, 3, 1, 1)
    x = (x - mean) / std
    th, tw = H // 14 * 14, W // 14 * 14
    t, l = int(round((H - th) / 2.0)), int(round((W - tw) / 2.0))
    ref = x[..., t:t + th, l:l + tw]
    assert out.shape == ref.shape
    assert torch.equal(out, ref)


@pytest.mark.parametrize("H,W,max_size", [(480, 640, 322), (700, 500, 448), (333, 1024, 1024), (2000, 1500, 1024)])
def test_ingest_with_bicubic_downscale(dev, H, W, max_size):
    """uint8 ingest with the demo's downscale rule (demo/anyloc_vlad_generate.py:163-181): ToTensor + Normalize, bicubic
    resize so the longer side is max_img_size (aspect kept), CenterCrop to multiples of 14 -- vs the same steps in torch."""
    from anyloc_amd import preprocess, synth
    g = torch.Generator().manual_seed(H * 3 + W)
    u8 = (torch.rand(2, H // 8, W // 8, 3, generator=g) * 255).to(torch.uint8)
    u8 = torch.nn.functional.interpolate(u8.permute(0, 3, 1, 2).float(), size=(H, W), mode="bilinear").permute(0, 2, 3, 1)
    u8 = (u8 + 8 * torch.rand(2, H, W, 3, generator=g)).clamp(0, 255).to(torch.uint8)
    out = preprocess.images_to_input(u8, max_img_size=max_size).cpu()
    x = u8.permute(0, 3, 1, 2).float().div(255)
    x = (x - torch.tensor(synth.IMAGENET_MEAN).view(1, 3, 1, 1)) / torch.tensor(synth.IMAGENET_STD).view(1, 3, 1, 1)
    if max(H, W) > max_size:
        if H == max(H, W):
            w2, h2 = int(W * max_size / H), max_size
        else:
            h2, w2 = int(H * max_size / W), max_size
        x = torch.nn.functional.interpolate(x, size=(h2, w2), mode="bicubic", align_corners=False)
    h, w = x.shape[-2:]
    th, tw = h // 14 * 14, w // 14 * 14
    t, l = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
    ref = x[..., t:t + th, l:l + tw]
    assert out.shape == ref.shape and out.shape[-1] % 14 == 0 and out.shape[-2] % 14 == 0
    assert float((out - ref).abs().max()) < 2e-5
    # the resize alone, an upscale too
    y = torch.randn(1, 3, 37, 53, generator=g)
    up = preprocess.resize_bicubic(y.to(dev), (90, 61)).cpu()
    assert float((up - torch.nn.functional.interpolate(y, size=(90, 61), mode="bicubic", align_corners=False)).abs().max()) < 1e-5
