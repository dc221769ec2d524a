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


def _centre(full, part):
    return int(round((full - part) / 2.0))          # torchvision CenterCrop: round half to even


def resize_bicubic(x, size, crop=None):
    """x float [B,3,H,W] (device) -> bicubic resize to ``size=(h,w)`` (torch's kernel, align_corners=False, no
    antialias = ``T.resize(img, (h, w), BICUBIC)`` of a tensor in the torchvision the reference pins), optionally
    centre-cropped to ``crop=(ch,cw)`` in the same kernel."""
    dev = _lib.require_gpu()
    x = x.to(dev, torch.float32).contiguous()
    B, Cn, H, W = x.shape
    h, w = int(size[0]), int(size[1])
    ch, cw = (h, w) if crop is None else (int(crop[0]), int(crop[1]))
    out = torch.empty(B, Cn, ch, cw, dtype=torch.float32, device=dev)
    _lib.check(_lib.load().anyloc_resize_bicubic(_lib.ptr(x), B * Cn, H, W, h, w, _centre(h, ch), _centre(w, cw), ch, cw,
                                                 _lib.ptr(out), _lib.stream_ptr()), "anyloc_resize_bicubic")
    return out


def images_to_input(images, mean=IMAGENET_MEAN, std=IMAGENET_STD, multiple=PATCH, crop=None, max_img_size=None):
    """images: uint8 [B,H,W,3] (or [H,W,3]) torch tensor / numpy array, host or device.
    Returns float32 [B,3,H',W'] on the GPU with H' = H//multiple*multiple (or ``crop=(h,w)``).

    ``max_img_size`` (reference demo/anyloc_vlad_generate.py:163-181): when the longer side exceeds it, the normalised
    image is first resized (aspect kept, ``int()`` truncation as the demo computes it) with bicubic interpolation, then
    centre-cropped to multiples of ``multiple`` -- normalise, resize and crop all on the device."""
    dev = _lib.require_gpu()
    if isinstance(images, np.ndarray):
        images = torch.from_numpy(np.ascontiguousarray(images))
    if images.ndim == 3:
        images = images[None]
    if images.dtype != torch.uint8 or images.shape[-1] != 3:
        raise ValueError(f"expected uint8 [B,H,W,3], got {images.dtype} {tuple(images.shape)}")
    images = images.to(dev, non_blocking=True).contiguous()
    B, H, W, _ = images.shape
    if max_img_size is not None and max(H, W) > max_img_size:
        if H == max(H, W):
            w2, h2 = int(W * max_img_size / H), int(max_img_size)
        else:
            h2, w2 = int(H * max_img_size / W), int(max_img_size)
        full = images_to_input(images, mean, std, multiple, crop=(H, W))          # ToTensor + Normalize, no crop
        tgt = crop if crop is not None else (h2 // multiple * multiple, w2 // multiple * multiple)
        return resize_bicubic(full, (h2, w2), tgt)
    ch, cw = crop if crop is not None else (H // multiple * multiple, W // multiple * multiple)
    out = torch.empty(B, 3, ch, cw, dtype=torch.float32, device=dev)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    _lib.check(_lib.load().anyloc_preprocess_u8(C.c_void_p(images.data_ptr()), B, H, W, ch, cw, m, s,
                                                _lib.ptr(out), _lib.stream_ptr()), "anyloc_preprocess_u8")
    return out
