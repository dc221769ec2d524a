"""Device-side ingest: uint8 HWC images -> ImageNet-normalised float CHW
batches cropped to multiples of the patch size, in one HIP kernel -- the work the reference does per
image on the host with ``ToTensor() + Normalize`` (``dvgl_benchmark/datasets_ws.py:20-23``) and
``CenterCrop((h//14*14, w//14*14))`` (``scripts/dino_v2_vlad.py:173-176``) before every extractor call.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .synth import IMAGENET_MEAN, IMAGENET_STD, PATCH


def images_to_input(images, mean=IMAGENET_MEAN, std=IMAGENET_STD, multiple=PATCH, crop=None):
    """images: uint8 [B,H,W,3] (or [H,W,3]) torch tensor / numpy array, host or device.
    Returns float32 [B,3,H',W'] on the GPU with H' = H//multiple*multiple (or ``crop=(h,w)``)."""
    dev = _lib.require_gpu()
    if isinstance(images, np.ndarray):
        images = torch.from_numpy(np.ascontiguousarray(images))
    if images.ndim == 3:
        images = images[None]
    if images.dtype != torch.uint8 or images.shape[-1] != 3:
        raise ValueError(f"expected uint8 [B,H,W,3], got {images.dtype} {tuple(images.shape)}")
    images = images.to(dev, non_blocking=True).contiguous()
    B, H, W, _ = images.shape
    ch, cw = crop if crop is not None else (H // multiple * multiple, W // multiple * multiple)
    out = torch.empty(B, 3, ch, cw, dtype=torch.float32, device=dev)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(_lib.load().anyloc_preprocess_u8(C.c_void_p(images.data_ptr()), B, H, W, ch, cw, m, s,
                                                _lib.ptr(out), _lib.stream_ptr()), "anyloc_preprocess_u8")
    return out
