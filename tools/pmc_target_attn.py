"""rocprofv3 --pmc target: the h3 attention kernel alone at the bench shape (B=61, T=530, 24 heads)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyloc_amd import ops  # noqa: E402

qkv = torch.randn(61, 530, 3 * 1536, device="cuda")
for _ in range(5):
    img, inv = ops.attention_h3(qkv, 24)
torch.cuda.synchronize()
print("ok")
