"""Attention variants (ANYLOC_OPTIONS=attn_cfg=<n>): time + error vs fp64 on the ViT-g shape."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops  # noqa: E402

dev = "cuda"
B, T, H = 61, 530, 24
g = torch.Generator(device=dev)
g.manual_seed(0)
qkv = torch.randn(B, T, 3 * H * 64, generator=g, device=dev) * 1.5
for _ in range(3):
    out = ops.attention(qkv, H)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = ops.attention(qkv, H)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
q, k, v = qkv[:4].double().reshape(4, T, 3, H, 64).permute(2, 0, 3, 1, 4)
ref = (torch.softmax((q * 0.125) @ k.transpose(-2, -1), dim=-1) @ v).transpose(1, 2).reshape(4, T, H * 64)
err = float((out[:4].double() - ref).abs().max())
rel = float((out[:4].double() - ref).norm() / ref.norm())
print(f"options[{os.environ.get('ANYLOC_OPTIONS', '')}]: {dt*1e3:7.3f} ms  {4.0*B*H*T*T*64/dt/1e12:6.1f} TF/s  max abs err {err:.2e}  rel {rel:.2e}", flush=True)
