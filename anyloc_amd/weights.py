"""Where DINOv2 weights come from.

The reference calls ``torch.hub.load('facebookresearch/dinov2', name)``
(``utilities.py:239-240``), which downloads code and a checkpoint.  This
implementation needs only the *state dict* (hub key names).  Resolution order:

1. a state dict registered in-process with :func:`register_state_dict`
   (tests / benchmarks inject seeded synthetic weights this way);
2. ``$ANYLOC_DINOV2_WEIGHTS`` -- a ``.pth`` file, or a directory holding
   ``<name>_pretrain.pth`` (the file names facebookresearch publishes);
3. the torch hub checkpoint cache (``torch.hub.get_dir()/checkpoints``), where
   a previous ``torch.hub.load`` of the real model would have left it;
4. ``$ANYLOC_SYNTHETIC_WEIGHTS=<seed>`` -- seeded random weights (explicit opt-in);
5. download with ``torch.hub.load_state_dict_from_url`` (needs network).
"""
import os

import torch

from .synth import ARCH, synthetic_state_dict

_REGISTERED = {}
_URL = "https://dl.fbaipublicfiles.com/dinov2/{short}/{short}_pretrain.pth"


def register_state_dict(name, state_dict):
    _REGISTERED[name] = state_dict


def unregister_state_dict(name=None):
    if name is None:
        _REGISTERED.clear()
    else:
        _REGISTERED.pop(name, None)


def resolve_state_dict(name):
    if name not in ARCH:
        raise ValueError(f"unknown DINOv2 model {name!r}; expected one of {sorted(ARCH)}")
    if name in _REGISTERED:
        return _REGISTERED[name]
    fname = f"{name}_pretrain.pth"
    cands = []
    env = os.environ.get("ANYLOC_DINOV2_WEIGHTS")
    if env:
        cands.append(env if os.path.isfile(env) else os.path.join(env, fname))
    cands.append(os.path.join(torch.hub.get_dir(), "checkpoints", fname))
    for c in cands:
        if os.path.isfile(c):
            return torch.load(c, map_location="cpu")
    seed = os.environ.get("ANYLOC_SYNTHETIC_WEIGHTS")
    if seed is not None:
        print(f"[anyloc_amd] using SYNTHETIC {name} weights (seed {seed})")
        return synthetic_state_dict(name, int(seed))
    try:
        return torch.hub.load_state_dict_from_url(_URL.format(short=name), map_location="cpu")
    except Exception as exc:   # no network
        raise FileNotFoundError(
            f"no weights for {name}: set ANYLOC_DINOV2_WEIGHTS to a checkpoint / directory, place "
            f"{fname} in {os.path.join(torch.hub.get_dir(), 'checkpoints')}, or opt in to random "
            f"weights with ANYLOC_SYNTHETIC_WEIGHTS=<seed> (download failed: {exc})") from exc
