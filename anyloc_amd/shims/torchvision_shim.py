"""Import-time stand-in for the slice of ``torchvision.transforms`` the reference's dataset
loaders and the two target scripts touch (``dvgl_benchmark/datasets_ws.py:13,20-30,236-257``,
``scripts/dino_v2_vlad.py:32,176``, ``demo/anyloc_vlad_generate.py:133-137,175-181``).
Data preparation only -- nothing here is on the accelerated path."""
import types

import numpy as np
import torch
from PIL import Image
from torch.nn import functional as F


class InterpolationMode:
    NEAREST, BILINEAR, BICUBIC = "nearest", "bilinear", "bicubic"


_PIL = {"nearest": Image.NEAREST, "bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC}


def _size2(size, h, w):
    if isinstance(size, int):                      # shorter side -> size, keep aspect
        if h <= w:
            return size, max(1, int(size * w / h))
        return max(1, int(size * h / w)), size
    if len(size) == 1:
        return _size2(int(size[0]), h, w)
    return int(size[0]), int(size[1])


def resize(img, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=None):
    if isinstance(img, Image.Image):
        h, w = _size2(size, img.height, img.width)
        return img.resize((w, h), _PIL[interpolation])
    h, w = _size2(size, img.shape[-2], img.shape[-1])
    lead = img.shape[:-3]
    x = img.reshape(-1, *img.shape[-3:]).float()
    kw = {} if interpolation == "nearest" else dict(align_corners=False, antialias=bool(antialias))
    x = F.interpolate(x, size=(h, w), mode=interpolation, **kw)
    return x.reshape(*lead, *x.shape[-3:]).to(img.dtype if img.is_floating_point() else torch.float32)


def center_crop(img, output_size):
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    th, tw = int(output_size[0]), int(output_size[1])
    if isinstance(img, Image.Image):
        w, h = img.size
        l, t = int(round((w - tw) / 2.0)), int(round((h - th) / 2.0))
        return img.crop((l, t, l + tw, t + th))
    h, w = img.shape[-2:]
    t, l = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
    return img[..., t:t + th, l:l + tw]


def five_crop(img, size):
    if isinstance(size, int):
        size = (size, size)
    th, tw = size
    h, w = img.shape[-2:]
    return (img[..., :th, :tw], img[..., :th, w - tw:], img[..., h - th:, :tw], img[..., h - th:, w - tw:],
            center_crop(img, size))


def to_tensor(pic):
    if isinstance(pic, torch.Tensor):
        return pic
    arr = np.array(pic)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
    return t.float().div(255) if t.dtype == torch.uint8 else t.float()


def normalize(t, mean, std):
    mean = torch.as_tensor(mean, dtype=t.dtype, device=t.device).view(-1, 1, 1)
    std = torch.as_tensor(std, dtype=t.dtype, device=t.device).view(-1, 1, 1)
    return (t - mean) / std


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    def __call__(self, pic):
        return to_tensor(pic)


class Normalize:
    def __init__(self, mean, std, inplace=False):
        self.mean, self.std = mean, std

    def __call__(self, t):
        return normalize(t, self.mean, self.std)


class CenterCrop:
    def __init__(self, size):
        self.size = size

    def __call__(self, img):
        return center_crop(img, self.size)


class Resize:
    def __init__(self, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=None):
        self.size, self.interpolation, self.antialias = size, interpolation, antialias

    def __call__(self, img):
        return resize(img, self.size, self.interpolation, antialias=self.antialias)


class Lambda:
    def __init__(self, lambd):
        self.lambd = lambd

    def __call__(self, x):
        return self.lambd(x)


def _training_only(name):
    class _T:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            raise NotImplementedError(f"torchvision.transforms.{name} (training augmentation) is not provided")
    _T.__name__ = name
    return _T


def build_modules():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    fn = types.ModuleType("torchvision.transforms.functional")
    for k, v in dict(resize=resize, center_crop=center_crop, five_crop=five_crop, to_tensor=to_tensor,
                     normalize=normalize, InterpolationMode=InterpolationMode).items():
        setattr(fn, k, v)
    for k, v in dict(Compose=Compose, ToTensor=ToTensor, Normalize=Normalize, CenterCrop=CenterCrop, Resize=Resize,
                     Lambda=Lambda, InterpolationMode=InterpolationMode, functional=fn).items():
        setattr(tr, k, v)
    for name in ("ColorJitter", "RandomPerspective", "RandomResizedCrop", "RandomRotation", "RandomHorizontalFlip"):
        setattr(tr, name, _training_only(name))
    tv.transforms = tr
    tv.__version__ = "0.0-anyloc-shim"
    return {"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms.functional": fn}
