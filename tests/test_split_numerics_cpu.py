"""CPU pins of the arithmetic the split GEMM kernels are built on (csrc/gemm_x6.hip, csrc/gemm_h3.hip), emulated
with torch on the host exactly as tools/split_bf16_study.py / tools/split_fp16_study.py do:
  * three bf16 planes reproduce an fp32 number exactly, and the six leading plane products give a GEMM that is as
    accurate as an fp32 GEMM (the three-product variant is not -- the reason the kernel issues six);
  * two fp16 planes of a row scaled into [2^14, 2^15) carry 22 bits relative to the row maximum without overflow,
    and the three products hh + hl + lh give an fp32-accurate GEMM after descaling.
The kernels themselves are checked on the GPU (tests/test_gpu_x6.py); this file keeps the numerical design honest
on machines without one."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import split_bf16_study as sb  # noqa: E402
import split_fp16_study as sh  # noqa: E402


def operands(seed, M=192, N=160, K=768):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g) * (0.25 + torch.rand(M, 1, generator=g))
    a[:, ::37] *= 40.0                                   # heavy-tailed columns inside every row
    w = torch.randn(N, K, generator=g) * 0.03
    return a, w


def rel_err(c, a, w):
    ref = a.double() @ w.double().t()
    mag = a.double().abs() @ w.double().abs().t()
    return float(((c.double() - ref).abs() / mag).max())


def test_three_bf16_planes_are_exact():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * torch.exp(6 * torch.randn(4096, generator=g))
    x[0], x[1] = 0.0, 1.0 + 2.0 ** -23
    p1, p2, p3 = sb.split3(x)
    assert torch.equal(p1.double() + p2.double() + p3.double(), x.double())
    assert torch.equal(p1, x.bfloat16().float())


@pytest.mark.parametrize("seed", [1, 2])
def test_six_bf16_products_match_fp32_accuracy_three_do_not(seed):
    a, w = operands(seed)
    e32 = rel_err(a @ w.t(), a, w)
    e6 = rel_err(sb.make_linear(sb.T6)(a, w), a, w)
    e3 = rel_err(sb.make_linear(sb.T3)(a, w), a, w)
    assert e6 < 1.5 * e32 + 1e-7, (e6, e32)
    assert e3 > 4 * e6, (e3, e6)                          # 2^-16-level terms are missing


def test_fp16_row_scaled_split_has_22_bits_and_no_overflow():
    a, _ = operands(3)
    a[7] = 0.0
    a[8] *= 1e-20
    a[9] *= 1e20
    h, l, scale = sh.split_h2(a)
    amax = a.abs().amax(dim=1, keepdim=True)
    scaled = amax * scale
    ok = amax.squeeze(1) > 0
    assert bool(((scaled[ok] >= 2.0 ** 14) & (scaled[ok] < 2.0 ** 15)).all())
    assert float(h.abs().max()) < 65504 and torch.isfinite(l).all()
    back = (h.double() + l.double()) / scale.double()
    assert float(((back - a.double()).abs() / amax.double().clamp_min(1e-300)).max()) < 2.0 ** -22
    assert float(back[7].abs().max()) == 0.0


@pytest.mark.parametrize("seed", [4, 5])
def test_three_fp16_products_match_fp32_accuracy(seed):
    a, w = operands(seed)
    e32 = rel_err(a @ w.t(), a, w)
    e3 = rel_err(sh.linear_h3(torch.nn.functional.linear)(a, w), a, w)
    assert e3 < 1.5 * e32 + 1e-7, (e3, e32)
