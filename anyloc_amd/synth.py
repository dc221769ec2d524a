"""Deterministic synthetic weights / images / vocabularies.

There is no network in the build or GPU environment, so neither the DINOv2
checkpoints nor the VPR datasets the reference uses are available (SURVEY.md
section 7 "Hard parts").  Benchmarks and parity tests therefore run on
seeded synthetic inputs of the exact shapes the reference would see.  The
state dicts use the facebookresearch/dinov2 hub key names, so a real
``dinov2_vit*14_pretrain.pth`` is interchangeable with them.
"""
import math

import torch

# name -> (embed dim, depth, heads, ffn kind, ffn hidden)   [hub backbones]
ARCH = {
    "dinov2_vits14": (384, 12, 6, "mlp", 1536),
    "dinov2_vitb14": (768, 12, 12, "mlp", 3072),
    "dinov2_vitl14": (1024, 24, 16, "mlp", 4096),
    "dinov2_vitg14": (1536, 40, 24, "swiglu", 4096),
}
PATCH = 14
POS_GRID = 37
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def _randn(gen, shape, std, device):
    t = torch.empty(shape, dtype=torch.float32, device=device)
    t.normal_(0.0, std, generator=gen)
    return t.clamp_(-2.5 * std, 2.5 * std)


def synthetic_state_dict(name, seed=0, device="cpu", depth=None):
    """Random-init weights with the hub model's key names and shapes.

    Not an identity-ish init: biases, LayerNorm affine parameters and LayerScale
    gammas are all non-trivial so that every fused epilogue is exercised.
    ``depth`` truncates the block list (tests only need blocks <= hook layer).
    """
    dim, full_depth, heads, ffn, hidden = ARCH[name]
    depth = full_depth if depth is None else depth
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    r = lambda shape, std: _randn(gen, shape, std, device)
    sd = {}
    sd["cls_token"] = r((1, 1, dim), 0.02)
    sd["mask_token"] = torch.zeros(1, dim, device=device)
    # smooth positional table: low-frequency random field + small noise, so
    # that bicubic interpolation of it is a meaningful operation
    coarse = r((1, dim, 6, 6), 0.05)
    grid = torch.nn.functional.interpolate(coarse, size=(POS_GRID, POS_GRID),
                                           mode="bilinear", align_corners=True)
    grid = grid.permute(0, 2, 3, 1).reshape(1, POS_GRID * POS_GRID, dim)
    grid = grid + r((1, POS_GRID * POS_GRID, dim), 0.005)
    sd["pos_embed"] = torch.cat([r((1, 1, dim), 0.02), grid], dim=1).contiguous()
    sd["patch_embed.proj.weight"] = r((dim, 3, PATCH, PATCH), 0.03)
    sd["patch_embed.proj.bias"] = r((dim,), 0.02)
    w_std = 0.7 / math.sqrt(dim)
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = 1.0 + r((dim,), 0.1)
        sd[p + "norm1.bias"] = r((dim,), 0.05)
        sd[p + "attn.qkv.weight"] = r((3 * dim, dim), 2.0 * w_std)
        sd[p + "attn.qkv.bias"] = r((3 * dim,), 0.05)
        sd[p + "attn.proj.weight"] = r((dim, dim), w_std)
        sd[p + "attn.proj.bias"] = r((dim,), 0.02)
        sd[p + "ls1.gamma"] = 0.3 + 0.2 * torch.rand(dim, generator=gen, device=device)
        sd[p + "norm2.weight"] = 1.0 + r((dim,), 0.1)
        sd[p + "norm2.bias"] = r((dim,), 0.05)
        if ffn == "mlp":
            sd[p + "mlp.fc1.weight"] = r((hidden, dim), w_std)
            sd[p + "mlp.fc1.bias"] = r((hidden,), 0.05)
            sd[p + "mlp.fc2.weight"] = r((dim, hidden), 0.7 / math.sqrt(hidden))
            sd[p + "mlp.fc2.bias"] = r((dim,), 0.02)
        else:
            sd[p + "mlp.w12.weight"] = r((2 * hidden, dim), w_std)
            sd[p + "mlp.w12.bias"] = r((2 * hidden,), 0.05)
            sd[p + "mlp.w3.weight"] = r((dim, hidden), 0.7 / math.sqrt(hidden))
            sd[p + "mlp.w3.bias"] = r((dim,), 0.02)
        sd[p + "ls2.gamma"] = 0.3 + 0.2 * torch.rand(dim, generator=gen, device=device)
    sd["norm.weight"] = 1.0 + r((dim,), 0.1)
    sd["norm.bias"] = r((dim,), 0.05)
    return sd


def synthetic_places(n_db, n_qu, h, w, seed=42, device="cpu"):
    """Synthetic place-recognition set: ``n_db`` "places" (low-pass random RGB
    fields) and ``n_qu`` queries, query q = place (q mod n_db) + pixel noise +
    a small circular shift.  Returns ImageNet-normalised float32 ``[n,3,h,w]``
    tensors (what ``ToTensor()+Normalize`` hands the reference's extractor,
    reference ``dvgl_benchmark/datasets_ws.py:20-23``) and ``gt_pos`` (object
    array, ``gt_pos[q] = [q mod n_db]``)."""
    import numpy as np
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    ch, cw = max(h // 8, 2), max(w // 8, 2)
    coarse = torch.rand(n_db, 3, ch, cw, generator=gen, device=device)
    db = torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear",
                                         align_corners=False)
    db = (0.9 * db + 0.1 * torch.rand(n_db, 3, h, w, generator=gen, device=device))
    qu = []
    for q in range(n_qu):
        img = db[q % n_db]
        sh = int(torch.randint(0, 8, (1,), generator=gen, device=device))
        sw = int(torch.randint(0, 8, (1,), generator=gen, device=device))
        img = torch.roll(img, shifts=(sh, sw), dims=(1, 2))
        img = img + 0.05 * torch.randn(3, h, w, generator=gen, device=device)
        qu.append(img.clamp(0, 1))
    qu = torch.stack(qu) if n_qu else torch.empty(0, 3, h, w, device=device)
    mean = torch.tensor(IMAGENET_MEAN, device=device).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=device).view(1, 3, 1, 1)
    gt = np.empty(n_qu, dtype=object)
    for q in range(n_qu):
        gt[q] = np.array([q % n_db])
    return ((db - mean) / std).contiguous(), ((qu - mean) / std).contiguous(), gt


def clustered_tokens(n_img, n_tok, dim, n_modes, seed=7, device="cpu", noise=0.35):
    """Unit-norm tokens drawn around ``n_modes`` random directions (mirrors
    real ViT features, whose top-2 cosine gaps are O(0.1)): ``[n_img,n_tok,dim]``."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    modes = torch.nn.functional.normalize(
        torch.randn(n_modes, dim, generator=gen, device=device), dim=1)
    pick = torch.randint(0, n_modes, (n_img, n_tok), generator=gen, device=device)
    x = modes[pick] + (noise / math.sqrt(dim)) * torch.randn(
        n_img, n_tok, dim, generator=gen, device=device)
    return torch.nn.functional.normalize(x, dim=-1)


def outlier_state_dict(sd, name, seed=0, gamma_boost=1000.0, resid_bias=600.0, cls_boost=100.0):
    """Numerical-stress variant of a hub-layout state dict: the statistics real DINOv2 checkpoints
    have and ``synthetic_state_dict`` does not (heavy-tailed LayerNorm gains, "massive activation"
    residual channels, LayerScale gammas spread over five decades, a register-like token).

      * LayerNorm gamma / beta of three channels x ``gamma_boost`` in every block, with the matching
        COLUMNS of the consuming projection (qkv, fc1 / w12) divided by the same factor: the function
        is preserved, but every GEMM operand row now spans three more decades (what a row-scaled
        fixed-point split has to survive);
      * ``blocks.0.attn.proj.bias`` of one channel = ``resid_bias``: a residual-stream channel that
        dominates every token's LayerNorm statistics from block 0 on;
      * a random third of the LayerScale gammas drawn log-uniformly from [1e-5, 1];
      * the CLS token x ``cls_boost`` (one token whose norm is far above the others').
    Returns a new dict; ``sd`` is not modified."""
    dim = ARCH[name][0]
    ffn = ARCH[name][3]
    out = {k: v.clone() for k, v in sd.items()}
    dev = out["cls_token"].device
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    hot = torch.randperm(dim, generator=gen)[:4].to(dev)
    depth = 1 + max(int(k.split(".")[1]) for k in out if k.startswith("blocks."))
    fc1 = "mlp.fc1.weight" if ffn == "mlp" else "mlp.w12.weight"
    for i in range(depth):
        p = f"blocks.{i}."
        for norm, cons in (("norm1", "attn.qkv.weight"), ("norm2", fc1)):
            out[p + norm + ".weight"][hot[:3]] *= gamma_boost
            out[p + norm + ".bias"][hot[:3]] *= gamma_boost
            out[p + cons][:, hot[:3]] /= gamma_boost
        for ls in ("ls1.gamma", "ls2.gamma"):
            pick = (torch.rand(dim, generator=gen) < 1.0 / 3.0).to(dev)
            val = torch.pow(10.0, -5.0 * torch.rand(dim, generator=gen)).to(dev)
            out[p + ls] = torch.where(pick, val, out[p + ls])
    out["blocks.0.attn.proj.bias"][hot[3]] = resid_bias
    out["cls_token"] = out["cls_token"] * cls_boost
    return out
