"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's non-VLAD aggregations.

The reference defines these inline inside script-local closures (they cannot be imported), so the
restatement follows the lines themselves:

* ``global_pool``      -- scripts/dino_v2_gp.py:130-135  (mean / max over the token axis)
* ``gem_descriptors``  -- scripts/dino_v2_gem.py:170-188 (three GeM variants, incl. the complex64 root)
* ``cls_descriptor``   -- scripts/dino_v2_global_vpr.py:115-128 (hub model forward = final LayerNorm of
                          the CLS token, identity head)

Parity: unpinned by upstream tests (the reference has none); the functions below ARE the reference's
torch expressions, evaluated on the CPU.
"""
import torch


def global_pool(patch_descs: torch.Tensor, pool_method: str = "average") -> torch.Tensor:
    """[n_img, N, D] -> [n_img, D]."""
    if pool_method == "average":
        return torch.mean(patch_descs, dim=1)
    if pool_method == "max":
        return torch.max(patch_descs, dim=1)[0]
    raise NotImplementedError(f"ID: {pool_method}")


def gem_descriptors(patch_descs: torch.Tensor, gem_p: float = 3, gem_use_abs: bool = False,
                    gem_elem_by_elem: bool = False) -> torch.Tensor:
    """[n_img, N, D] -> [n_img, D]; ``gem_elem_by_elem`` only changes the loop structure upstream."""
    assert patch_descs.dim() == 3
    if gem_use_abs:
        return torch.mean(torch.abs(patch_descs) ** gem_p, dim=-2) ** (1 / gem_p)
    x = torch.mean(patch_descs ** gem_p, dim=-2)
    root = x.to(torch.complex64) ** (1 / gem_p)       # complex root: |x|^(1/p) e^{i pi/p} for x < 0
    return torch.abs(root) * torch.sign(x)


def cls_descriptor(model, img: torch.Tensor) -> torch.Tensor:
    """Hub-model forward of the oracle DINOv2 (all blocks, final norm, CLS row): [B,3,H,W] -> [B,D]."""
    with torch.no_grad():
        return model(img)
