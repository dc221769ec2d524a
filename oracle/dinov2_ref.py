"""TEST INFRASTRUCTURE -- pure-torch (CPU, fp32) restatement of the DINOv2 ViT.

The reference obtains the backbone with ``torch.hub.load('facebookresearch/
dinov2', name)`` (``utilities.py:239-240``, no ref pinned => ``main``), i.e. the
model code is a third-party dependency that is NOT under ``/root/reference``
and cannot be fetched here (no network).  Its published architecture is
restated below, following ``dinov2/models/vision_transformer.py``
(``DinoVisionTransformer``, ``interpolate_pos_encoding`` with
``interpolate_offset=0.1``, ``prepare_tokens_with_masks``),
``dinov2/layers/{patch_embed,attention,block,mlp,swiglu_ffn,layer_scale}.py``
and ``dinov2/hub/backbones.py`` (img_size 518, patch 14, init_values 1.0,
ffn "mlp" for S/B/L and "swiglufused" for g, block_chunks 0, no registers).

Sub-module names equal the hub model's, so (a) real ``dinov2_vit*14_pretrain
.pth`` state dicts load with ``load_state_dict`` unchanged, and (b) the
reference's forward hooks on ``blocks[L].attn.qkv`` / ``blocks[L]``
(``utilities.py:245-252``) attach to this model exactly as to the hub one.
Independent cross-check: ``tests/test_oracle_dinov2_hf.py`` compares it with
``transformers.Dinov2Model`` through a q/k/v weight remap.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

# name -> (embed dim, depth, heads, ffn kind, ffn hidden)
ARCH = {
    "dinov2_vits14": (384, 12, 6, "mlp", 1536),
    "dinov2_vitb14": (768, 12, 12, "mlp", 3072),
    "dinov2_vitl14": (1024, 24, 16, "mlp", 4096),
    # SwiGLUFFNFused: hidden = (int(4*D*2/3) + 7) // 8 * 8 = 4096 for D=1536
    "dinov2_vitg14": (1536, 40, 24, "swiglu", 4096),
}
PATCH = 14
POS_GRID = 37            # 518 / 14
INTERP_OFFSET = 0.1


class _PatchEmbed(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, PATCH, PATCH)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)      # [B, N, D]


class _Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, T, D = x.shape
        hd = D // self.heads
        qkv = self.qkv(x).reshape(B, T, 3, self.heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
        a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((a @ v).transpose(1, 2).reshape(B, T, D))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))                 # exact (erf) GELU


class _SwiGLU(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.w12 = nn.Linear(dim, 2 * hidden)
        self.w3 = nn.Linear(hidden, dim)

    def forward(self, x):
        x1, x2 = self.w12(x).chunk(2, dim=-1)
        return self.w3(F.silu(x1) * x2)


class _LayerScale(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _Block(nn.Module):
    def __init__(self, dim, heads, ffn, hidden):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, heads)
        self.ls1 = _LayerScale(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, hidden) if ffn == "mlp" else _SwiGLU(dim, hidden)
        self.ls2 = _LayerScale(dim)

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


def interpolate_pos_embed(pos_embed, h_img, w_img):
    """[1, 1+37*37, D] table -> [1, 1+(h/14)*(w/14), D] for an h x w image.
    Bicubic, align_corners=False, no antialias, driven by scale_factor with the
    +0.1 offset (NOT by output size); skipped for the native square grid."""
    n_tab = pos_embed.shape[1] - 1
    gh, gw = h_img // PATCH, w_img // PATCH
    if gh * gw == n_tab and h_img == w_img:
        return pos_embed
    m = int(math.sqrt(n_tab))
    assert m * m == n_tab
    dim = pos_embed.shape[-1]
    grid = pos_embed[:, 1:].float().reshape(1, m, m, dim).permute(0, 3, 1, 2)
    sf = (float(gh + INTERP_OFFSET) / m, float(gw + INTERP_OFFSET) / m)
    grid = F.interpolate(grid, scale_factor=sf, mode="bicubic", antialias=False)
    assert tuple(grid.shape[-2:]) == (gh, gw)
    grid = grid.permute(0, 2, 3, 1).reshape(1, gh * gw, dim)
    return torch.cat([pos_embed[:, :1].float(), grid], dim=1).to(pos_embed.dtype)


class DinoVisionTransformer(nn.Module):
    def __init__(self, name):
        super().__init__()
        dim, depth, heads, ffn, hidden = ARCH[name]
        self.name, self.embed_dim, self.depth, self.num_heads = name, dim, depth, heads
        self.patch_embed = _PatchEmbed(dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + POS_GRID * POS_GRID, dim))
        self.mask_token = nn.Parameter(torch.zeros(1, dim))
        self.blocks = nn.ModuleList(_Block(dim, heads, ffn, hidden) for _ in range(depth))
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.head = nn.Identity()

    def prepare_tokens(self, img):
        B, _, H, W = img.shape
        x = self.patch_embed(img)
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], dim=1)
        return x + interpolate_pos_embed(self.pos_embed, H, W)

    def forward(self, img, n_blocks=None):
        """Full forward as the reference runs it (all blocks, final norm, head
        on CLS).  ``n_blocks`` limits the depth (used only to prove that the
        product's early exit is equivalent for the hooked layer)."""
        x = self.prepare_tokens(img)
        for blk in self.blocks[:n_blocks]:
            x = blk(x)
        return self.head(self.norm(x)[:, 0])


def build(name, state_dict):
    model = DinoVisionTransformer(name)
    model.load_state_dict(state_dict, strict=True)
    return model.eval()


@torch.no_grad()
def extract_facet(model, img, layer, facet="value", use_cls=False, norm_descs=True):
    """Restatement of ``DinoV2ExtractFeatures.__call__`` (reference
    ``utilities.py:263-285``) on top of a model built by :func:`build`:
    hook output of ``blocks[layer].attn.qkv`` (q/k/v) or ``blocks[layer]``
    (token), CLS dropped unless ``use_cls``, facet = third of the last dim,
    ``F.normalize(dim=-1)``."""
    grabbed = {}
    target = model.blocks[layer] if facet == "token" else model.blocks[layer].attn.qkv
    handle = target.register_forward_hook(lambda m, i, o: grabbed.__setitem__("o", o))
    try:
        model(img)
    finally:
        handle.remove()
    res = grabbed["o"]
    if not use_cls:
        res = res[:, 1:, ...]
    if facet in ("query", "key", "value"):
        d = res.shape[2] // 3
        j = ("query", "key", "value").index(facet)
        res = res[:, :, j * d:(j + 1) * d]
    if norm_descs:
        res = F.normalize(res, dim=-1)
    return res
