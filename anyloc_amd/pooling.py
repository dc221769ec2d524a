"""Non-VLAD aggregations of DINOv2 patch tokens on the device.

The reference computes these with torch expressions inside its scripts
(``scripts/dino_v2_gp.py:130-135``, ``scripts/dino_v2_gem.py:170-188``); here each is one pass of the
HBM-bound ``anyloc_pool_tokens`` kernel over tokens that already live on the GPU.  Inputs on the CPU are
staged to the GPU and the result returns on the input's device, like the rest of the drop-in surface.
"""
import torch

from . import ops


def _back(res, like):
    return res if (torch.is_tensor(like) and like.is_cuda) or not torch.is_tensor(like) else res.to(like.device)


def global_pool(patch_descs, pool_method: str = "average") -> torch.Tensor:
    """[n_img, N, D] (or a list of [N_i, D]) -> [n_img, D]; ``pool_method`` in {"average", "max"}
    (anything else raises NotImplementedError, as ``scripts/dino_v2_gp.py:134-135``)."""
    if pool_method not in ("average", "max"):
        raise NotImplementedError(f"ID: {pool_method}")
    return _back(ops.pool(patch_descs, pool_method), patch_descs)


def gem_descriptors(patch_descs, gem_p: float = 3, gem_use_abs: bool = False,
                    gem_elem_by_elem: bool = False) -> torch.Tensor:
    """GeM pooling with the reference's three switches (``scripts/dino_v2_gem.py:88-107,170-188``):
    ``gem_use_abs`` -> mean(|t|^p)^(1/p); otherwise |mean(t^p)|^(1/p) * sign(mean(t^p)) (the modulus of the
    complex root the reference takes).  ``gem_elem_by_elem`` only changes the reference's Python loop."""
    if torch.is_tensor(patch_descs):
        assert patch_descs.dim() == 3, "expected [N, n_p, d_dim]"
    return _back(ops.pool(patch_descs, "gem_abs" if gem_use_abs else "gem", gem_p), patch_descs)
